/* compatibility shim: the reference splits its API over several headers (include/agmv_playback.h);
   this build keeps one. */
#ifndef AGMV_SHIM_agmv_playback
#define AGMV_SHIM_agmv_playback
#include "agmv.h"
#endif
