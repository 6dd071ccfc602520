/* TEST INFRASTRUCTURE ONLY: __half for the host emulation lives in the emulated hip_runtime.h */
#include <hip/hip_runtime.h>
