// respmon_amd/csrc/rm_unity.hip -- every translation unit of the library as ONE unit.  Used by the tracing developer build (its
// device-side trace buffer is one global) and by the host emulation of the tests (tests/emu/build.py); the product library is
// linked from the separate units (Makefile).
#include "rm_ctx.hip"
#include "rm_pyramid.hip"
#include "rm_temporal.hip"
#include "rm_down.hip"
#include "rm_down_f64.hip"
#include "rm_down_generic.hip"
#include "rm_down_narrow.hip"
#include "rm_down_bgr.hip"
#include "rm_front.hip"
#include "rm_collapse_eval.hip"
#include "rm_collapse_sum.hip"
#include "rm_calibrate.hip"
#include "rm_roi.hip"
#include "rm_locate.hip"
#include "rm_comm.hip"
#include "rm_motion.hip"
