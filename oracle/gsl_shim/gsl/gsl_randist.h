/* oracle/_ref build only: see gsl_cdf.h */
#pragma once
#include "gsl_cdf.h"
