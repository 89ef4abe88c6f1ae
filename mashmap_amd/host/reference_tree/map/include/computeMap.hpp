// Drop-in replacement for src/map/include/computeMap.hpp inside the reference tree:
// skch::Map backed by libmashmap_hip.so.
#pragma once
#ifndef MASHMAP_HIP_REFERENCE_TREE
#define MASHMAP_HIP_REFERENCE_TREE 1
#endif
#include "skch_map.hpp"
