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
  for (int i = tid; i < 17000; i += 256) lds[i] = __builtin_sinf((float)i * 12.9898f) * 43758.5453f;
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
  } else if (MODE == 20 || MODE == 21) {   // MODE 17 with other MFMA shapes: 20 v_mfma_f32_4x4x1 (2 passes), 21 v_mfma_f32_32x32x2 (16 passes)
    const float a = (float)lane * 0.37f, b = 1.0f + lane;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 d0 = {0.f}, d1 = {0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        __builtin_amdgcn_s_sleep(2);
        if (MODE == 20) {
          c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
        } else {
          d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, d1, 0, 0, 0);
        }
      }
    }
    acc = c0[0] + c1[1] + d0[0] + d1[5];
  } else if (MODE == 17 || MODE == 18 || MODE == 19) {   // SPARSE MFMAs without any LDS: 17 an s_sleep (~128 clocks) between pairs,
    const float a = (float)lane * 0.37f, b = 1.0f + lane;   // 18 a dependent VALU chain (~40 FMAs) between pairs, 19 bursts of 16 then a long sleep
    float x = a;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (MODE == 17) __builtin_amdgcn_s_sleep(2);
        if (MODE == 18) {
#pragma unroll
          for (int j = 0; j < 40; ++j) x = __builtin_fmaf(x, 1.0001f, 0.25f);
        }
        if (MODE == 19 && (k & 7) == 0) __builtin_amdgcn_s_sleep(16);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE == 18 ? x : a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, MODE == 18 ? x : a, c1, 0, 0, 0);
      }
    }
    acc = c0[0] + c1[1] + x;
  } else if (MODE == 15 || MODE == 16) {   // as 8, but the gathered value passes through ONE VALU instruction before the MFMAs read it
    float bw[32][2];                       // (15: v_mov_b32; 16: v_add_f32 with 0)
#pragma unroll
    for (int k = 0; k < 32; ++k) { bw[k][0] = lds[(lane * 3 + k * 17) & 8191] * 1.37f; bw[k][1] = lds[(lane * 5 + k * 29) & 8191] * 0.73f; }
    const float* base = lds + (lane & 15) * 2 + (lane >> 4) + (tid >> 6) * 1100;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float g = base[(k >> 2) * 625 + (k & 3) * 25 + (i & 7)];
        float a;
        if (MODE == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(g));
        else asm volatile("v_add_f32 %0, 0, %1" : "=v"(a) : "v"(g));
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][1], c1, 0, 0, 0);
      }
    }
    acc = c0[0] + c1[1] + c0[2] + c1[3];
  } else if (MODE >= 12 && MODE <= 14) {   // 12: the gather feeds two VALU FMAs instead of MFMAs; 13: gathers and MFMAs interleaved but
                                           // INDEPENDENT (the MFMAs take a register operand); 14: as 8 with the gather two steps ahead
    float bw[32][2];
#pragma unroll
    for (int k = 0; k < 32; ++k) { bw[k][0] = lds[(lane * 3 + k * 17) & 8191] * 1.37f; bw[k][1] = lds[(lane * 5 + k * 29) & 8191] * 0.73f; }
    const float* base = lds + (lane & 15) * 2 + (lane >> 4) + (tid >> 6) * 1100;
    float areg = bw[0][0] * 0.5f, x0 = 0.f, x1 = 0.f, side = 0.f;
    for (int i = 0; i < iters; ++i) {
      float pre0 = base[(i & 7)], pre1 = base[25 + (i & 7)];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (MODE == 12) {
          const float a = base[(k >> 2) * 625 + (k & 3) * 25 + (i & 7)];
          x0 = __builtin_fmaf(a, bw[k][0], x0);
          x1 = __builtin_fmaf(a, bw[k][1], x1);
        } else if (MODE == 13) {
          side += base[(k >> 2) * 625 + (k & 3) * 25 + (i & 7)];
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg, bw[k][0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg, bw[k][1], c1, 0, 0, 0);
        } else {
          const float a = pre0;
          pre0 = pre1;
          pre1 = base[(((k + 2) & 31) >> 2) * 625 + ((k + 2) & 3) * 25 + (i & 7)];
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][1], c1, 0, 0, 0);
        }
      }
      areg = areg * 1.0001f + 0.3f;
    }
    acc = c0[0] + c1[1] + c0[2] + c1[3] + x0 + x1 + side;
  } else if (MODE >= 9 && MODE <= 11) {   // MODE 8 minus one ingredient: 9 no LDS gather in the loop, 10 quiet data, 11 one operand register pair
    float bw[32][2];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int kk = MODE == 11 ? 0 : k;
      bw[k][0] = MODE == 10 ? 1.0f : lds[(lane * 3 + kk * 17) & 8191] * 1.37f;
      bw[k][1] = MODE == 10 ? 1.0f : lds[(lane * 5 + kk * 29) & 8191] * 0.73f;
    }
    if (MODE == 10) { __syncthreads(); for (int i = tid; i < 17000; i += 256) lds[i] = 1.0f; __syncthreads(); }
    const float* base = lds + (lane & 15) * 2 + (lane >> 4) + (tid >> 6) * 1100;
    float areg = bw[0][0] * 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float a = MODE == 9 ? areg : base[(k >> 2) * 625 + (k & 3) * 25 + (i & 7)];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][1], c1, 0, 0, 0);
      }
      if (MODE == 9) areg = areg * 1.0001f + 0.3f;
    }
    acc = c0[0] + c1[1] + c0[2] + c1[3];
  } else if (MODE == 8) {   // conv2's inner loop in miniature: an LDS gather per pair of MFMAs, 64 B operands in registers, noisy data
    float bw[32][2];
#pragma unroll
    for (int k = 0; k < 32; ++k) { bw[k][0] = lds[(lane * 3 + k * 17) & 8191] * 1.37f; bw[k][1] = lds[(lane * 5 + k * 29) & 8191] * 0.73f; }
    const float* base = lds + (lane & 15) * 2 + (lane >> 4) + (tid >> 6) * 1100;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float a = base[(k >> 2) * 625 + (k & 3) * 25 + (i & 7)];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[k][1], c1, 0, 0, 0);
      }
    }
    acc = c0[0] + c1[1] + c0[2] + c1[3];
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
    hipFuncSetAttribute((const void*)neighbour_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<21>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<19>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<18>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<15>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)neighbour_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
  else if (mode == 8) neighbour_kernel<8><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 21) neighbour_kernel<21><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 20) neighbour_kernel<20><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 19) neighbour_kernel<19><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 18) neighbour_kernel<18><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 17) neighbour_kernel<17><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 16) neighbour_kernel<16><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 15) neighbour_kernel<15><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 14) neighbour_kernel<14><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 13) neighbour_kernel<13><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 12) neighbour_kernel<12><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 11) neighbour_kernel<11><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 10) neighbour_kernel<10><<<grid, 256, lds, s>>>(sink, iters);
  else if (mode == 9) neighbour_kernel<9><<<grid, 256, lds, s>>>(sink, iters);
  else neighbour_kernel<3><<<grid, 256, lds, s>>>(sink, iters);
  return (int)hipGetLastError();
}
