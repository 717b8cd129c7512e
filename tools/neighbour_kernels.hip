// Dev tool (GPU box): synthetic NEIGHBOURS for the emulator — kernels that occupy a CU the way the learner's conv
// kernels do (512 workgroups of 256 threads, ~70 KB of LDS each: two per CU, ~230 VGPRs) but stress ONE resource each,
// to find out which one stretches atari_env_kernel beside them (tools/env_beside_neighbours.py).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/neighbour_kernels.hip -o build_exp/neighbours.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: LDS reads (strided, like conv12's gathers)  1: MFMA chains  2: VALU FMAs  3: resident, asleep
__global__ __launch_bounds__(256, 2) void neighbour_kernel(float* sink, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 17000; i += 256) lds[i] = (float)i;
  __syncthreads();
  float acc = 0.f;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
  if (MODE == 0) {
    const float* base = lds + (lane & 15) * 2 + (lane >> 4) + (tid >> 6) * 1100;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float a = base[(k >> 2) * 1936 + (k & 3) * 44], b = base[(k >> 2) * 1936 + (k & 3) * 44 + 625];
        acc += a + b;
      }
      asm volatile("" : "+v"(acc));
    }
  } else if (MODE == 1) {
    const float a = (float)lane, b = 1.0f + lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
      }
    }
    acc = c0[0] + c1[1];
  } else if (MODE == 2) {
    float x = (float)lane, y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) { x = __builtin_fmaf(x, y, 0.5f); acc = __builtin_fmaf(acc, y, x); }
    }
  } else if (MODE == 6) {   // 16 KB of straight-line code per pass (2,000 8-byte VALU instructions), four waves at different offsets
    float x = (float)lane;
    for (int i = 0; i < iters; ++i) {
      asm volatile(".rept 2000\n v_fma_f32 %0, %0, 1.0, 0.5\n .endr" : "+v"(x));
    }
    acc = x;
  } else if (MODE == 7) {   // 48 KB of straight-line code per pass
    float x = (float)lane;
    for (int i = 0; i < iters; ++i) {
      asm volatile(".rept 6000\n v_fma_f32 %0, %0, 1.0, 0.5\n .endr" : "+v"(x));
    }
    acc = x;
  } else if (MODE == 4) {   // VALU FMAs over ~200 live registers (the conv kernels' register footprint, nothing else of them)
    float r[200];
#pragma unroll
    for (int k = 0; k < 200; ++k) r[k] = lds[(lane + k * 7) & 1023];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 200; ++k) r[k] = __builtin_fmaf(r[k], 1.0001f, 0.5f);
#pragma unroll
      for (int k = 0; k < 200; ++k) asm volatile("" : "+v"(r[k]));
    }
#pragma unroll
    for (int k = 0; k < 200; ++k) acc += r[k];
  } else if (MODE == 5) {   // the same registers held, asleep
    float r[200];
#pragma unroll
    for (int k = 0; k < 200; ++k) r[k] = lds[(lane + k * 7) & 1023];
#pragma unroll
    for (int k = 0; k < 200; ++k) asm volatile("" : "+v"(r[k]));
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
#pragma unroll
    for (int k = 0; k < 200; ++k) asm volatile("" : "+v"(r[k]));
#pragma unroll
    for (int k = 0; k < 200; ++k) acc += r[k];
  } else {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
  }
  if (acc == 123.456f) sink[tid] = acc;
}

extern "C" int neighbour_launch(int mode, float* sink, int iters, int grid, void* stream) {
  const size_t lds = 70000;
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute((const void*)neighbour_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) neighbour_kernel<0><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 1) neighbour_kernel<1><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 2) neighbour_kernel<2><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 4) neighbour_kernel<4><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 5) neighbour_kernel<5><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 6) neighbour_kernel<6><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 7) neighbour_kernel<7><<<grid, 256, lds, s>>>(sink, iters);
  else neighbour_kernel<3><<<grid, 256, lds, s>>>(sink, iters);
  return (int)hipGetLastError();
}
