/* compatibility shim: the reference splits its API over several headers (include/agmv_defines.h);
   this build keeps one. */
#ifndef AGMV_SHIM_agmv_defines
#define AGMV_SHIM_agmv_defines
#include "agmv.h"
#endif
