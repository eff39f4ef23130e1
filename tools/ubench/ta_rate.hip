// Vector-memory issue rate per CU when everything hits L1 / L2: how many cycles does the texture addresser + L1 spend per wave instruction
// for the access shapes of k_project_gather?   hipcc --offload-arch=gfx950 -O3 ta_rate.hip -o ta_rate && ./ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: dwordx4, 8 lanes per 128-B line, 8 lines per instruction taken from a small table (L1 hits)
// mode 1: dword, fully coalesced 256 B per instruction
// mode 2: dwordx4 fully coalesced 1 KiB per instruction
// mode 3: dwordx3 (12 B) per lane, every lane its own line
// mode 4: dwordx4, 8 lanes per line, lines spread over 2.4 MB per view x 8 views (L2 / MALL hits)
template <int MODE>
__global__ void __launch_bounds__(256) k_load(const float4* __restrict__ buf, long n4, int iters, float* __restrict__ out, unsigned seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  unsigned s = seed + blockIdx.x * 977u + wave * 131u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 1664525u + 1013904223u;
      long idx;
      if (MODE == 0) idx = (long)(((s >> 8) + (lane >> 3) * 37) & 63) * 8 + (lane & 7);
      else if (MODE == 1) idx = 0;
      else if (MODE == 2) idx = (long)((s >> 8) & 15) * 64 + lane;
      else if (MODE == 3) idx = (long)(((s >> 8) + lane * 97) & 8191) * 8;
      else idx = ((long)(((s >> 6) + (lane >> 3) * 7919u) % (unsigned)(n4 / 8))) * 8 + (lane & 7);
      if (MODE == 1) {
        acc += reinterpret_cast<const float*>(buf)[(((s >> 8) & 63) * 64) + lane];
      } else if (MODE == 3) {
        const float* p = reinterpret_cast<const float*>(buf + idx);
        typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));
        const f3 v = *reinterpret_cast<const f3*>(p + 1);
        acc += v.x + v.y + v.z;
      } else {
        const float4 v = buf[idx];
        acc += v.x + v.y + v.z + v.w;
      }
    }
  }
  if (acc == 1.2345f) out[0] = acc;
}

// stores: dwordx4, fully coalesced, streaming (each wave its own 1 KiB per instruction) over a buffer of `span4` float4:
// mode 0 plain, 1 non-temporal builtin, 2 "sc1", 3 "sc0 sc1", 4 "nt sc1", 5 "nt sc0 sc1" (cache-policy bits of global_store on gfx942/950)
template <int MODE>
__global__ void __launch_bounds__(256) k_store(float4* __restrict__ buf, long span4, int iters) {
  const long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (long)iters * 8 * 64 + (threadIdx.x & 63);
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 t = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f4* p = reinterpret_cast<f4*>(buf + (base + (long)(it * 8 + u) * 64) % span4);
      if (MODE == 0) *p = t;
      else if (MODE == 1) __builtin_nontemporal_store(t, p);
      else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
      else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
      else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(t) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off nt sc0 sc1" ::"v"(p), "v"(t) : "memory");
    }
}

int main() {
  const long n4 = 8L * 2400 * 1024 / 16;  // 8 "views" of 2.4 MB
  float4* buf; float* out;
  CHECK(hipMalloc(&buf, n4 * 16)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 0, n4 * 16));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 200;
  const char* names[5] = {"dwordx4 8 lanes/line, L1-resident table", "dword coalesced 256 B", "dwordx4 coalesced 1 KiB", "dwordx3 one line per lane", "dwordx4 8 lanes/line over 19 MB"};
  for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
    const int grid = cus * wg_per_cu;
    for (int mode = 0; mode < 5; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        switch (mode) {
          case 0: k_load<0><<<grid, 256>>>(buf, n4, iters, out, 1); break;
          case 1: k_load<1><<<grid, 256>>>(buf, n4, iters, out, 1); break;
          case 2: k_load<2><<<grid, 256>>>(buf, n4, iters, out, 1); break;
          case 3: k_load<3><<<grid, 256>>>(buf, n4, iters, out, 1); break;
          default: k_load<4><<<grid, 256>>>(buf, n4, iters, out, 1); break;
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double instr_per_cu = (double)wg_per_cu * 4 * iters * 8;
      printf("load  %-44s %d waves/CU: %7.1f us  %6.1f cycles per wave-instruction per CU (at %.2f GHz)\n", names[mode], wg_per_cu * 4, ms * 1e3,
             ms * 1e-3 * ghz * 1e9 / instr_per_cu, ghz);
    }
  }
  // streaming stores over a buffer far larger than every cache (1.34 GB) and over a cache-resident one (19 MB)
  float4* big; const long big4 = 1340L * 1024 * 1024 / 16;
  CHECK(hipMalloc(&big, big4 * 16));
  const char* snames[6] = {"plain", "nt (builtin)", "sc1", "sc0 sc1", "nt sc1", "nt sc0 sc1"};
  for (int which = 0; which < 2; ++which)
    for (int mode = 0; mode < 6; ++mode) {
      const int wg_per_cu = 4, grid = cus * wg_per_cu;
      float4* dst = which ? buf : big; const long span = which ? n4 : big4;
      const int it2 = which ? iters : 160;  // 1024 WGs x 4 waves x 160 x 8 KiB = 5.4 GB = four sweeps of the big buffer
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        switch (mode) {
          case 0: k_store<0><<<grid, 256>>>(dst, span, it2); break;
          case 1: k_store<1><<<grid, 256>>>(dst, span, it2); break;
          case 2: k_store<2><<<grid, 256>>>(dst, span, it2); break;
          case 3: k_store<3><<<grid, 256>>>(dst, span, it2); break;
          case 4: k_store<4><<<grid, 256>>>(dst, span, it2); break;
          default: k_store<5><<<grid, 256>>>(dst, span, it2); break;
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double bytes = (double)grid * 4 * it2 * 8 * 1024;
      printf("store dwordx4 1 KiB %-14s %s: %8.1f us  %6.1f cycles per wave-instruction per CU, %.2f TB/s\n", snames[mode], which ? "19 MB buffer  " : "1.34 GB buffer",
             ms * 1e3, ms * 1e-3 * ghz * 1e9 / ((double)wg_per_cu * 4 * it2 * 8), bytes / (ms * 1e-3) / 1e12);
    }
  return 0;
}
