// <cuda_fp16.h> stand-in (test infrastructure, see cuda_runtime_api.h): nv_util.h includes it, the
// cache's float path uses nothing of it
#pragma once
