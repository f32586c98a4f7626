// tools/micro/bank.hip -- does the VGPR bank (register number mod 4) of the sources change the issue cost of a VALU
// instruction on gfx950? Explicit register numbers, 8 independent chains, 2 waves per SIMD (both alive for the whole
// launch), cycles per wave-instruction per SIMD from the waves' own s_memtime span.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bank tools/micro/bank.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
             "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131"
// chains i = 0..7: destination / first source v(100+i); the other sources chosen per case
#define INIT asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 1.0\n v_mov_b32 v102, 1.0\n v_mov_b32 v103, 1.0\n v_mov_b32 v104, 1.0\n v_mov_b32 v105, 1.0\n v_mov_b32 v106, 1.0\n v_mov_b32 v107, 1.0\n" \
  "v_mov_b32 v108, 1.0\n v_mov_b32 v109, 1.0\n v_mov_b32 v110, 1.0\n v_mov_b32 v111, 1.0\n v_mov_b32 v112, 1.0\n v_mov_b32 v113, 1.0\n v_mov_b32 v114, 1.0\n v_mov_b32 v115, 1.0\n" \
  "v_mov_b32 v116, 1.0\n v_mov_b32 v117, 1.0\n v_mov_b32 v118, 1.0\n v_mov_b32 v119, 1.0\n v_mov_b32 v120, 1.0\n v_mov_b32 v121, 1.0\n v_mov_b32 v122, 1.0\n v_mov_b32 v123, 1.0\n" ::: CLOB)

enum { MUL_SAME, MUL_DIFF, FMA_ALL_DIFF, FMA_TWO_SAME, FMA_ALL_SAME, ADD_SAME, ADD_DIFF, MUL_DST_OTHER, NCASE };
static const char* NAMES[] = {"v_mul_f32  src0, src1 in the SAME bank", "v_mul_f32  src0, src1 in DIFFERENT banks", "v_fma_f32  three different banks",
  "v_fma_f32  src1, src2 same bank, src0 another", "v_fma_f32  all three in one bank", "v_add_f32  same bank", "v_add_f32  different banks",
  "v_mul_f32  different banks, dst in a third bank"};

template <int C>
__global__ __launch_bounds__(256) void k(unsigned long long* span, int iters) {
  INIT;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (C == MUL_SAME) {          // v(100+i) *= v(108+i): 100+i and 108+i are congruent mod 4
      for (int u = 0; u < 4; ++u)
        asm volatile("v_mul_f32 v100, v100, v108\n v_mul_f32 v101, v101, v109\n v_mul_f32 v102, v102, v110\n v_mul_f32 v103, v103, v111\n"
                     "v_mul_f32 v104, v104, v112\n v_mul_f32 v105, v105, v113\n v_mul_f32 v106, v106, v114\n v_mul_f32 v107, v107, v115" ::: CLOB);
    } else if (C == MUL_DIFF) {   // v(100+i) *= v(109+i): banks differ by 1
      for (int u = 0; u < 4; ++u)
        asm volatile("v_mul_f32 v100, v100, v109\n v_mul_f32 v101, v101, v110\n v_mul_f32 v102, v102, v111\n v_mul_f32 v103, v103, v112\n"
                     "v_mul_f32 v104, v104, v113\n v_mul_f32 v105, v105, v114\n v_mul_f32 v106, v106, v115\n v_mul_f32 v107, v107, v116" ::: CLOB);
    } else if (C == FMA_ALL_DIFF) {  // v(100+i) = v(100+i) * v(109+i) + v(118+i): banks i, i+1, i+2
      for (int u = 0; u < 4; ++u)
        asm volatile("v_fma_f32 v100, v100, v109, v118\n v_fma_f32 v101, v101, v110, v119\n v_fma_f32 v102, v102, v111, v120\n v_fma_f32 v103, v103, v112, v121\n"
                     "v_fma_f32 v104, v104, v113, v122\n v_fma_f32 v105, v105, v114, v123\n v_fma_f32 v106, v106, v115, v116\n v_fma_f32 v107, v107, v108, v117" ::: CLOB);
    } else if (C == FMA_TWO_SAME) {  // src1, src2 same bank: v(109+i), v(117+i)
      for (int u = 0; u < 4; ++u)
        asm volatile("v_fma_f32 v100, v100, v109, v117\n v_fma_f32 v101, v101, v110, v118\n v_fma_f32 v102, v102, v111, v119\n v_fma_f32 v103, v103, v112, v120\n"
                     "v_fma_f32 v104, v104, v113, v121\n v_fma_f32 v105, v105, v114, v122\n v_fma_f32 v106, v106, v115, v123\n v_fma_f32 v107, v107, v108, v116" ::: CLOB);
    } else if (C == FMA_ALL_SAME) {  // v(100+i), v(108+i), v(116+i)
      for (int u = 0; u < 4; ++u)
        asm volatile("v_fma_f32 v100, v100, v108, v116\n v_fma_f32 v101, v101, v109, v117\n v_fma_f32 v102, v102, v110, v118\n v_fma_f32 v103, v103, v111, v119\n"
                     "v_fma_f32 v104, v104, v112, v120\n v_fma_f32 v105, v105, v113, v121\n v_fma_f32 v106, v106, v114, v122\n v_fma_f32 v107, v107, v115, v123" ::: CLOB);
    } else if (C == ADD_SAME) {
      for (int u = 0; u < 4; ++u)
        asm volatile("v_add_f32 v100, v100, v108\n v_add_f32 v101, v101, v109\n v_add_f32 v102, v102, v110\n v_add_f32 v103, v103, v111\n"
                     "v_add_f32 v104, v104, v112\n v_add_f32 v105, v105, v113\n v_add_f32 v106, v106, v114\n v_add_f32 v107, v107, v115" ::: CLOB);
    } else if (C == ADD_DIFF) {
      for (int u = 0; u < 4; ++u)
        asm volatile("v_add_f32 v100, v100, v109\n v_add_f32 v101, v101, v110\n v_add_f32 v102, v102, v111\n v_add_f32 v103, v103, v112\n"
                     "v_add_f32 v104, v104, v113\n v_add_f32 v105, v105, v114\n v_add_f32 v106, v106, v115\n v_add_f32 v107, v107, v116" ::: CLOB);
    } else if (C == MUL_DST_OTHER) {  // v(124+i)... dst v(126+i) (bank i+2) = v(100+i) * v(109+i)
      for (int u = 0; u < 4; ++u)
        asm volatile("v_mul_f32 v126, v100, v109\n v_mul_f32 v127, v101, v110\n v_mul_f32 v128, v102, v111\n v_mul_f32 v129, v103, v112\n"
                     "v_mul_f32 v130, v104, v113\n v_mul_f32 v131, v105, v114\n v_mul_f32 v124, v106, v115\n v_mul_f32 v125, v107, v116" ::: CLOB);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) span[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int C> void run(int n_cu) {
  const int W = 2, blocks = n_cu * W, iters = 20000;
  unsigned long long* span; CHECK(hipMalloc(&span, 8 * blocks * 4));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(256), 0, 0, span, iters); CHECK(hipDeviceSynchronize()); }
  std::vector<unsigned long long> h(blocks * 4);
  CHECK(hipMemcpy(h.data(), span, 8 * blocks * 4, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-52s %5.2f cycles / wave-instruction / SIMD (2 waves per SIMD)\n", NAMES[C], (double)h[h.size() / 2] / ((double)iters * 32 * W));
  CHECK(hipFree(span));
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int n = p.multiProcessorCount;
  run<MUL_SAME>(n); run<MUL_DIFF>(n); run<MUL_DST_OTHER>(n); run<ADD_SAME>(n); run<ADD_DIFF>(n); run<FMA_ALL_DIFF>(n); run<FMA_TWO_SAME>(n); run<FMA_ALL_SAME>(n);
  return 0;
}
