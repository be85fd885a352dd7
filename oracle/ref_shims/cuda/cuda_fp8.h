// <cuda_fp8.h> stand-in (test infrastructure, see cuda_runtime_api.h): the two type names
// nv_util.h's is_fp8 trait is specialised for
#pragma once
struct __nv_fp8_e4m3 {
  unsigned char x;
};
struct __nv_fp8_e5m2 {
  unsigned char x;
};
