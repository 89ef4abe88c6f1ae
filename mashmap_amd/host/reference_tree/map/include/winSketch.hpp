// Drop-in replacement for src/map/include/winSketch.hpp inside the reference tree:
// skch::Sketch backed by libmashmap_hip.so.  Put this directory first on the include path and define
// MASHMAP_HIP_REFERENCE_TREE; mash_map.cpp and parseCmdArgs.hpp stay untouched.
#pragma once
#ifndef MASHMAP_HIP_REFERENCE_TREE
#define MASHMAP_HIP_REFERENCE_TREE 1
#endif
#include "skch_sketch.hpp"
