// hipemu.h -- TEST INFRASTRUCTURE ONLY: a lane-by-lane interpreter of the HIP execution model on
// the host, so that the kernels' own source (hugectr_amd/csrc/*.hip, compiled unchanged as C++
// against tests/emu/include/hip/*.h) can be stepped through on a machine without a GPU and its
// results checked against the oracle BEFORE GPU time is spent on it.  It is not a fallback: the
// product (hugectr_amd/) never builds, loads or links anything in this directory, and only tests/
// does.  What it checks is kernel LOGIC (indexing, barriers placed where data crosses lanes,
// ranks / scans / masks); it knows nothing of the memory model, of timing or of occupancy.
//
// Model: a launch runs its workgroups on OS threads (one per workgroup while the grid is small, so
// kernels that spin on a grid barrier find all their workgroups alive); inside a workgroup every
// thread is a fiber.  A fiber runs until it reaches __syncthreads(), a wavefront collective
// (__ballot, __shfl*, __all, __any, readfirstlane, wave_barrier) or the end of the kernel; when all
// 64 lanes of a wavefront are waiting, the lanes that wait at the same collective call site
// exchange their values and go on.  Lanes therefore do NOT run in lockstep between collectives --
// code that leans on implicit lockstep without a wave barrier gives wrong answers here, which is
// the point.  A collective met by part of a wavefront only (divergent control flow) is executed
// with the lanes that met it, lowest call site first, and counted in stats().
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {

struct Ids {
  dim3 tid, bid, bdim, gdim;
};
Ids* ids();  // of the running fiber

enum Op {
  OP_WAVE_BARRIER,
  OP_BALLOT,
  OP_ALL,
  OP_ANY,
  OP_SHFL,
  OP_SHFL_UP,
  OP_SHFL_DOWN,
  OP_SHFL_XOR,
  OP_FIRST
};

void syncthreads();
uint64_t collective(Op op, uint64_t value, int arg, int width, int site);
void spin_pause();  // s_sleep inside a spin loop: lets everybody else run
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
// width of a wavefront (64; 32 when the code under the interpreter is written for 32-lane warps,
// e.g. the reference's CUDA sources behind oracle/_ref/libref_cache.so) and a cap on the OS threads
// of a launch (0 = one per workgroup while the grid is small; 1 = workgroups strictly one after the
// other in block order, which makes lock-protected code deterministic)
// a launch the device would refuse (a grid or block dimension of zero, more than 1024 threads per
// block: hipErrorInvalidConfiguration) runs nothing and is remembered; take_launch_error() hands
// the flag to the next hipGetLastError() of the calling thread's process and clears it
bool take_launch_error();
void set_wave_width(int lanes);
void set_max_workers(size_t n);
void* dyn_shared();

struct Stats {
  uint64_t launches, blocks, divergent_collectives, shfl_from_inactive;
};
Stats stats();
void reset_stats();

}  // namespace hipemu
