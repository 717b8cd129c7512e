// Dev tool (GPU box): single-wave issue/latency cost model for the scalar interpreter
// (how many cycles a lone wavefront per SIMD pays per SALU op, taken / not-taken branch,
// v_readlane round trip, LDS round trip).  hipcc --offload-arch=gfx950 tools/issue_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define ITERS 2000

template <int TEST>
__global__ void k(unsigned long long* out, int n, uint32_t* sink) {
  __shared__ uint32_t lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t acc = 0;
  for (int i = 0; i < n; ++i) {
    if (TEST == 0) {  // 64 dependent s_add
      asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n .endr" : "+s"(acc) : : "scc");
    } else if (TEST == 1) {  // 64 independent-ish s_add (4 chains)
      uint32_t a = acc, b = 1, c = 2, d = 3;
      asm volatile(".rept 16\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n .endr"
                   : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
      acc = a + b + c + d;
    } else if (TEST == 2) {  // 32 taken short branches (each skips one instruction)
      asm volatile(".rept 32\n s_branch 1\n s_add_u32 %0, %0, 7\n s_add_u32 %0, %0, 1\n .endr" : "+s"(acc) : : "scc");
    } else if (TEST == 3) {  // 32 not-taken conditional branches
      asm volatile("s_cmp_eq_u32 0, 1\n .rept 32\n s_cbranch_scc1 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n .endr" : "+s"(acc) : : "scc");
    } else if (TEST == 4) {  // 32 x (s_add -> v_mov from sgpr -> v_readlane -> s_add) round trips
      uint32_t v;
      asm volatile(".rept 32\n v_mov_b32 %1, %0\n s_nop 0\n v_readlane_b32 %0, %1, 3\n s_add_u32 %0, %0, 1\n .endr" : "+s"(acc), "=&v"(v) : : "scc");
    } else if (TEST == 5) {  // 32 x LDS round trips (address depends on previous value)
      uint32_t v;
      asm volatile(".rept 32\n s_and_b32 %0, %0, 0xfc\n v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1\n s_lshl_b32 %0, %0, 2\n .endr"
                   : "+s"(acc), "=&v"(v) : : "scc");
    } else if (TEST == 6) {  // 32 taken branches, each over 64 bytes (a different cache line / fetch window)
      asm volatile(".rept 32\n s_branch 16\n .rept 16\n s_add_u32 %0, %0, 7\n .endr\n s_add_u32 %0, %0, 1\n .endr" : "+s"(acc) : : "scc");
    } else if (TEST == 7) {  // 32 x (cmp + taken conditional branch over one instruction)
      asm volatile(".rept 32\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1\n s_add_u32 %0, %0, 7\n s_add_u32 %0, %0, 1\n .endr" : "+s"(acc) : : "scc");
    } else if (TEST == 8) {  // 64 dependent VALU ops
      uint32_t v = acc;
      asm volatile(".rept 64\n v_add_u32 %0, %0, 1\n .endr" : "+v"(v));
      acc = __builtin_amdgcn_readfirstlane(v);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0xdeadbeef) sink[0] = acc;
}

template <int TEST>
void run(const char* name, int per_iter, int blocks, int threads) {
  unsigned long long* d; uint32_t* s;
  hipMalloc(&d, blocks * 8); hipMalloc(&s, 4);
  k<TEST><<<blocks, threads>>>(d, 10, s);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<TEST><<<blocks, threads>>>(d, ITERS, s);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024]; hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (int i = 0; i < blocks; ++i) cyc += h[i]; cyc /= blocks;
  printf("%-44s waves/CU %d: %.2f counter-ticks/op  %.2f ns/op\n", name, threads / 64, cyc / ((double)ITERS * per_iter),
         ms * 1e6 / ((double)ITERS * per_iter));
  hipFree(d); hipFree(s);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  for (int threads : {64, 256, 1024}) {
    if (only >= 0) {
      switch (only) {
        case 0: run<0>("dependent s_add", 64, 256, threads); break;
        case 1: run<1>("4 independent chains s_add", 64, 256, threads); break;
        case 2: run<2>("taken s_branch (+1 skipped, +1 add)", 32, 256, threads); break;
        case 3: run<3>("not-taken s_cbranch (+2 adds)", 32, 256, threads); break;
        case 4: run<4>("s->v_mov->v_readlane->s_add round trip", 32, 256, threads); break;
        case 5: run<5>("LDS read round trip (6 instr)", 32, 256, threads); break;
        case 6: run<6>("taken s_branch over 64 B (+1 add)", 32, 256, threads); break;
        case 7: run<7>("s_cmp + taken s_cbranch (+1 add)", 32, 256, threads); break;
        case 8: run<8>("dependent v_add", 64, 256, threads); break;
      }
      continue;
    }
    run<0>("dependent s_add", 64, 256, threads);
    run<1>("4 independent chains s_add", 64, 256, threads);
    run<2>("taken s_branch (+1 skipped, +1 add)", 32, 256, threads);
    run<3>("not-taken s_cbranch (+2 adds)", 32, 256, threads);
    run<4>("s->v_mov->v_readlane->s_add round trip", 32, 256, threads);
    run<5>("LDS read round trip (6 instr)", 32, 256, threads);
    run<6>("taken s_branch over 64 B (+1 add)", 32, 256, threads);
    run<7>("s_cmp + taken s_cbranch (+1 add)", 32, 256, threads);
    run<8>("dependent v_add", 64, 256, threads);
  }
  return 0;
}
