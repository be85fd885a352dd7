// hipemu.cpp -- the fiber scheduler behind tests/emu/hipemu.h (test infrastructure only).
#include "hipemu.h"

#include <sched.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>

namespace hipemu {
namespace {

enum State { READY = 0, AT_BARRIER = 1, AT_COLLECTIVE = 2, DONE = 3 };

struct Fiber {
  ucontext_t ctx;
  Ids id;
  int state;
  Op op;
  uint64_t val, result;
  int arg, width, site;
};

struct Worker {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  const std::function<void()>* body = nullptr;
  Fiber* cur = nullptr;
  char* stacks = nullptr;
  size_t stack_bytes = 0;
  void* dyn = nullptr;
};

thread_local Worker* tl_worker = nullptr;

std::atomic<uint64_t> g_launches{0}, g_blocks{0}, g_divergent{0}, g_inactive{0};
std::atomic<int> g_launch_error{0};
int g_wave = 64;            // lanes of a wavefront (set_wave_width)
size_t g_worker_cap = 0;    // OS threads of a launch (set_max_workers; 0 = the default policy)

size_t env_size(const char* name, size_t dflt) {
  const char* v = getenv(name);
  return v ? (size_t)strtoull(v, nullptr, 10) : dflt;
}

void trampoline() {
  Worker* w = tl_worker;
  Fiber* f = w->cur;
  (*w->body)();
  f->state = DONE;
  // (uc_link brings the worker's scheduler back)
}

void yield_to_scheduler() {
  Worker* w = tl_worker;
  Fiber* f = w->cur;
  swapcontext(&f->ctx, &w->sched);
}

[[noreturn]] void die(const char* what, Worker* w) {
  fprintf(stderr, "hipemu: %s\n", what);
  if (w) {
    int c[4] = {0, 0, 0, 0};
    for (auto& f : w->fibers) c[f.state]++;
    fprintf(stderr, "  block (%u,%u,%u): ready %d, at barrier %d, at collective %d, done %d\n",
            w->fibers[0].id.bid.x, w->fibers[0].id.bid.y, w->fibers[0].id.bid.z, c[0], c[1], c[2],
            c[3]);
  }
  abort();
}

// the lanes of wavefront [lo, hi) that wait at the lowest call site exchange values and resume
bool resolve_wave(Worker* w, size_t lo, size_t hi) {
  int min_site = 0x7fffffff;
  bool other = false;
  for (size_t i = lo; i < hi; i++) {
    Fiber& f = w->fibers[i];
    if (f.state == READY) return false;  // somebody of this wavefront still runs (spinning)
    if (f.state == AT_COLLECTIVE && f.site < min_site) min_site = f.site;
    if (f.state == AT_BARRIER) other = true;
  }
  if (min_site == 0x7fffffff) return false;
  uint64_t active = 0, pred = 0;
  Op op = OP_WAVE_BARRIER;
  int first = -1;
  for (size_t i = lo; i < hi; i++) {
    Fiber& f = w->fibers[i];
    if (f.state != AT_COLLECTIVE) continue;
    if (f.site != min_site) {
      other = true;
      continue;
    }
    if (first < 0) {
      first = (int)(i - lo);
      op = f.op;
    } else if (f.op != op) {
      die("lanes of one wavefront wait at the same call site with different collectives", w);
    }
    active |= 1ull << (i - lo);
    if (f.val != 0) pred |= 1ull << (i - lo);
  }
  if (other) g_divergent.fetch_add(1, std::memory_order_relaxed);
  for (size_t i = lo; i < hi; i++) {
    Fiber& f = w->fibers[i];
    if (f.state != AT_COLLECTIVE || f.site != min_site) continue;
    const int self = (int)(i - lo);
    const int wd = f.width;
    int index = self;
    bool is_shfl = false;
    switch (op) {
      case OP_WAVE_BARRIER: f.result = 0; break;
      case OP_BALLOT: f.result = pred; break;
      case OP_ALL: f.result = (pred == active) ? 1 : 0; break;
      case OP_ANY: f.result = (pred != 0) ? 1 : 0; break;
      case OP_FIRST: f.result = w->fibers[lo + (size_t)first].val; break;
      case OP_SHFL:
        index = (f.arg + (self & ~(wd - 1))) & (g_wave - 1);
        is_shfl = true;
        break;
      case OP_SHFL_UP:
        index = self - f.arg;
        if (index < (self & ~(wd - 1))) index = self;
        is_shfl = true;
        break;
      case OP_SHFL_DOWN:
        index = self + f.arg;
        if ((self & (wd - 1)) + f.arg >= wd) index = self;
        is_shfl = true;
        break;
      case OP_SHFL_XOR:
        index = self ^ f.arg;
        if (index >= ((self + wd) & ~(wd - 1))) index = self;
        is_shfl = true;
        break;
    }
    if (is_shfl) {
      if (index >= 0 && index < g_wave && ((active >> index) & 1ull)) {
        f.result = w->fibers[lo + (size_t)index].val;
      } else {
        f.result = 0;  // (the hardware hands back whatever the idle lane's register holds)
        g_inactive.fetch_add(1, std::memory_order_relaxed);
      }
    }
  }
  for (size_t i = lo; i < hi; i++) {
    Fiber& f = w->fibers[i];
    if (f.state == AT_COLLECTIVE && f.site == min_site) f.state = READY;
  }
  return true;
}

void run_block(Worker* w, dim3 grid, dim3 block, unsigned linear_block) {
  const size_t n = (size_t)block.x * block.y * block.z;
  dim3 bid(linear_block % grid.x, (linear_block / grid.x) % grid.y,
           linear_block / (grid.x * grid.y));
  for (size_t i = 0; i < n; i++) {
    Fiber& f = w->fibers[i];
    f.id.tid = dim3((unsigned)(i % block.x), (unsigned)((i / block.x) % block.y),
                    (unsigned)(i / ((size_t)block.x * block.y)));
    f.id.bid = bid;
    f.id.bdim = block;
    f.id.gdim = grid;
    f.state = READY;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = w->stacks + i * w->stack_bytes;
    f.ctx.uc_stack.ss_size = w->stack_bytes;
    f.ctx.uc_link = &w->sched;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  size_t live = n;
  // HIPEMU_SCHED: the order in which the runnable threads of a workgroup are resumed -- 0 lane 0
  // first (default), 1 last lane first, 2 a new pseudo-random order at every pass.  Results that
  // change with it point at data crossing threads without a barrier.
  static const size_t sched = env_size("HIPEMU_SCHED", 0);
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = (uint32_t)(sched == 1 ? n - 1 - i : i);
  uint64_t rnd = 0x9E3779B97F4A7C15ull * (linear_block + 1);
  while (live > 0) {
    bool ran = false;
    if (sched == 2) {
      for (size_t i = n - 1; i > 0; i--) {
        rnd ^= rnd << 13;
        rnd ^= rnd >> 7;
        rnd ^= rnd << 17;
        std::swap(order[i], order[rnd % (i + 1)]);
      }
    }
    for (size_t oi = 0; oi < n; oi++) {
      const size_t i = order[oi];
      Fiber& f = w->fibers[i];
      if (f.state != READY) continue;
      w->cur = &f;
      swapcontext(&w->sched, &f.ctx);
      ran = true;
      if (f.state == DONE) live--;
    }
    if (live == 0) break;
    bool released = false;
    const size_t wv = (size_t)g_wave;
    for (size_t lo = 0; lo < n; lo += wv) released |= resolve_wave(w, lo, lo + wv < n ? lo + wv : n);
    if (!released) {
      size_t at_bar = 0, ready = 0;
      for (auto& f : w->fibers) {
        at_bar += f.state == AT_BARRIER;
        ready += f.state == READY;
      }
      if (at_bar == live) {
        for (auto& f : w->fibers)
          if (f.state == AT_BARRIER) f.state = READY;
        released = true;
      } else if (ready == 0) {
        die("deadlock: part of a workgroup waits at __syncthreads, the rest never gets there", w);
      }
    }
    if (!ran && !released) die("deadlock: nothing can run", w);
  }
}

}  // namespace

Ids* ids() { return &tl_worker->cur->id; }

void syncthreads() {
  tl_worker->cur->state = AT_BARRIER;
  yield_to_scheduler();
}

uint64_t collective(Op op, uint64_t value, int arg, int width, int site) {
  Fiber* f = tl_worker->cur;
  f->op = op;
  f->val = value;
  f->arg = arg;
  f->width = width;
  f->site = site;
  f->state = AT_COLLECTIVE;
  yield_to_scheduler();
  return f->result;
}

void spin_pause() {
  sched_yield();
  yield_to_scheduler();  // (still READY: resumed on the scheduler's next pass)
}

void* dyn_shared() { return tl_worker->dyn; }

bool take_launch_error() { return g_launch_error.exchange(0) != 0; }

void set_wave_width(int lanes) {
  if (lanes != 64 && lanes != 32) die("set_wave_width: 32 or 64 lanes", nullptr);
  g_wave = lanes;
}
void set_max_workers(size_t n) { g_worker_cap = n; }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  if (nblocks == 0 || nthreads == 0 || nthreads > 1024) {
    g_launch_error.store(1);  // hipErrorInvalidConfiguration on the device: nothing runs
    return;
  }
  g_launches.fetch_add(1, std::memory_order_relaxed);
  g_blocks.fetch_add(nblocks, std::memory_order_relaxed);
  static const size_t max_workers = env_size("HIPEMU_MAX_WORKERS", 128);
  static const size_t stack_kb = env_size("HIPEMU_STACK_KB", 128);
  size_t hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  // one OS thread per workgroup while the grid is small: every workgroup of a kernel that spins on
  // a grid barrier is alive.  (Larger grids: workgroups queue on `hw` threads -- such a kernel
  // would run into its spin limit, which the kernels under test report as an error.)
  size_t nworkers = nblocks <= max_workers ? nblocks : hw;
  if (g_worker_cap > 0 && nworkers > g_worker_cap) nworkers = g_worker_cap;
  std::atomic<size_t> next{0};
  auto work = [&]() {
    Worker w;
    w.body = &body;
    w.stack_bytes = stack_kb * 1024;
    w.stacks = (char*)mmap(nullptr, nthreads * w.stack_bytes, PROT_READ | PROT_WRITE,
                           MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (w.stacks == (char*)MAP_FAILED) die("mmap of the fiber stacks failed", nullptr);
    w.fibers.resize(nthreads);
    if (shmem > 0) w.dyn = aligned_alloc(64, (shmem + 63) / 64 * 64);
    tl_worker = &w;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      run_block(&w, grid, block, (unsigned)b);
    }
    tl_worker = nullptr;
    munmap(w.stacks, nthreads * w.stack_bytes);
    free(w.dyn);
  };
  if (nworkers == 1) {
    // (a fresh thread all the same: __shared__ variables are thread_local, a workgroup must not
    //  inherit the caller's)
    std::thread t(work);
    t.join();
    return;
  }
  std::vector<std::thread> ts;
  ts.reserve(nworkers);
  for (size_t i = 0; i < nworkers; i++) ts.emplace_back(work);
  for (auto& t : ts) t.join();
}

Stats stats() {
  Stats s;
  s.launches = g_launches.load();
  s.blocks = g_blocks.load();
  s.divergent_collectives = g_divergent.load();
  s.shfl_from_inactive = g_inactive.load();
  return s;
}

void reset_stats() {
  g_launches = 0;
  g_blocks = 0;
  g_divergent = 0;
  g_inactive = 0;
}

}  // namespace hipemu

extern "C" {
// read by the tests (ctypes)
void hipemu_stats(uint64_t* out4) {
  hipemu::Stats s = hipemu::stats();
  out4[0] = s.launches;
  out4[1] = s.blocks;
  out4[2] = s.divergent_collectives;
  out4[3] = s.shfl_from_inactive;
}
void hipemu_reset_stats(void) { hipemu::reset_stats(); }
}
