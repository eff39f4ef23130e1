// Runtime half of the HIP emulator (see hip/hip_runtime.h).  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

namespace emu {
thread_local ThreadCtx tc;
}
alignas(16) float4 dyn_smem[emu::LDS_BYTES / 16];
extern "C" char* emu_lds_base() { return reinterpret_cast<char*>(dyn_smem); }

namespace emu {
void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const int nthreads = block.x * block.y * block.z;
  if (nthreads > MAX_THREADS || shmem > LDS_BYTES) { fprintf(stderr, "emu: launch too large (%d thr, %zu B LDS)\n", nthreads, shmem); abort(); }
  const int nwaves = (nthreads + WAVE - 1) / WAVE;
  BlockCtx bc;
  bc.waves = std::vector<WaveCtx>(nwaves);
  pthread_barrier_init(&bc.bar, nullptr, nthreads);
  for (int w = 0; w < nwaves; ++w) {
    bc.waves[w].n = std::min(WAVE, nthreads - w * WAVE);
    pthread_barrier_init(&bc.waves[w].bar, nullptr, bc.waves[w].n);
  }
  const long nblocks = (long)grid.x * grid.y * grid.z;
  // persistent worker threads: each plays one threadIdx for every block in turn
  std::vector<std::thread> th;
  th.reserve(nthreads);
  for (int t = 0; t < nthreads; ++t) {
    th.emplace_back([&, t]() {
      ThreadCtx& c = tc;
      c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      c.bdim = block; c.gdim = grid;
      c.lane = t % WAVE; c.wave = t / WAVE;
      c.w = &bc.waves[c.wave]; c.b = &bc; c.phase = 0;
      for (long b = 0; b < nblocks; ++b) {
        c.bid = dim3(b % grid.x, (b / grid.x) % grid.y, b / ((long)grid.x * grid.y));
        // poison LDS between blocks so reads of uninitialised LDS show up as NaNs
        if (t == 0) memset(dyn_smem, 0xFF, shmem ? shmem : 0);
        pthread_barrier_wait(&bc.bar);
        body();
        pthread_barrier_wait(&bc.bar);
      }
    });
  }
  for (auto& x : th) x.join();
  pthread_barrier_destroy(&bc.bar);
  for (auto& w : bc.waves) pthread_barrier_destroy(&w.bar);
}
}  // namespace emu
