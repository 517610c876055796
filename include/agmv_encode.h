/* compatibility shim: the reference splits its API over several headers (include/agmv_encode.h);
   this build keeps one. */
#ifndef AGMV_SHIM_agmv_encode
#define AGMV_SHIM_agmv_encode
#include "agmv.h"
#endif
