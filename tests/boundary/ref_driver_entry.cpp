// tests/boundary/ref_driver_entry.cpp — C entry points for the REFERENCE'S OWN decode driver: this file is compiled together with
// jxlcoder/src/main/cpp/interop/JxlDecoding.cpp (unchanged, from where it lies in the reference tree) and linked against
// jxl_coder_amd/compat/libjxl.so + libjxl_threads.so (include/jxl_amd_libjxl.h), i.e. the reference's libjxl call sequence
// (JxlDecoding.cpp:46-171) runs against the secondary drop-in boundary and ends in the HIP kernels.  Test infrastructure; built in
// the build container only (tests/test_libjxl_abi.py), nothing of the reference is copied.
#include "interop/JxlDecoding.h"
#include <string.h>
#include "boundary_entry.inc"
