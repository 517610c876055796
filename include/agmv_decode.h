/* compatibility shim: the reference splits its API over several headers (include/agmv_decode.h);
   this build keeps one. */
#ifndef AGMV_SHIM_agmv_decode
#define AGMV_SHIM_agmv_decode
#include "agmv.h"
#endif
