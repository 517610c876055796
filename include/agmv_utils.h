/* compatibility shim: the reference splits its API over several headers (include/agmv_utils.h);
   this build keeps one. */
#ifndef AGMV_SHIM_agmv_utils
#define AGMV_SHIM_agmv_utils
#include "agmv.h"
#endif
