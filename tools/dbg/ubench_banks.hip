// Micro-benchmark (diagnostic, not shipped): issue cost of fp64 VALU ops as a function of how
// many VGPR-pair sources they read and which register banks those sit in (gfx950, one wave/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define KERNEL(NAME, ASM)                                                                   \
  __global__ void NAME(double* out, unsigned long long* ticks, int iters) {                 \
    asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n" \
                 "v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n" \
                 "v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n" \
                 "v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n" \
                 "v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n"                      \
                 "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"                      \
                 "s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0\n"             \
                 ::: "v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v25","v26","v27", \
                     "v30","v31","v32","v33","v34","v35","v36","v37","s20","s21","s22","s23");                      \
    unsigned long long t0 = __builtin_readcyclecounter();                                   \
    for (int it = 0; it < iters; ++it) {                                                    \
      asm volatile(ASM "\n" ASM "\n" ASM "\n" ASM ::: "v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v25","v26","v27", \
                     "v30","v31","v32","v33","v34","v35","v36","v37");                      \
    }                                                                                       \
    unsigned long long t1 = __builtin_readcyclecounter();                                   \
    double r; asm volatile("v_add_f64 %0, v[10:11], v[12:13]" : "=v"(r));                   \
    out[blockIdx.x * 64 + threadIdx.x] = r;                                                 \
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                      \
  }

// four independent destinations per group; sources in various places
KERNEL(fma_v_s_s,  "v_fma_f64 v[10:11], v[10:11], s[20:21], s[20:21]\n v_fma_f64 v[12:13], v[12:13], s[20:21], s[20:21]\n v_fma_f64 v[14:15], v[14:15], s[20:21], s[20:21]\n v_fma_f64 v[16:17], v[16:17], s[20:21], s[20:21]")
KERNEL(fma_v_v_s,  "v_fma_f64 v[10:11], v[10:11], v[20:21], s[22:23]\n v_fma_f64 v[12:13], v[12:13], v[22:23], s[22:23]\n v_fma_f64 v[14:15], v[14:15], v[24:25], s[22:23]\n v_fma_f64 v[16:17], v[16:17], v[26:27], s[22:23]")
KERNEL(fma_v_v_v,  "v_fma_f64 v[10:11], v[10:11], v[20:21], v[30:31]\n v_fma_f64 v[12:13], v[12:13], v[22:23], v[32:33]\n v_fma_f64 v[14:15], v[14:15], v[24:25], v[34:35]\n v_fma_f64 v[16:17], v[16:17], v[26:27], v[36:37]")
// fmac form: dst is src2 (y += g*dl)
KERNEL(fmac_vv,    "v_fmac_f64 v[10:11], v[20:21], v[30:31]\n v_fmac_f64 v[12:13], v[22:23], v[32:33]\n v_fmac_f64 v[14:15], v[24:25], v[34:35]\n v_fmac_f64 v[16:17], v[26:27], v[36:37]")
// staggered banks: pairs start at 10 (bank 2), 21 -> use 20/22 vs 30/32: try to spread: src pairs at mod4 = 2,0,2 vs 2,0,0
KERNEL(fma_banks_a, "v_fma_f64 v[10:11], v[10:11], v[20:21], v[24:25]\n v_fma_f64 v[12:13], v[12:13], v[22:23], v[26:27]\n v_fma_f64 v[14:15], v[14:15], v[20:21], v[24:25]\n v_fma_f64 v[16:17], v[16:17], v[22:23], v[26:27]")
KERNEL(fma_banks_b, "v_fma_f64 v[10:11], v[10:11], v[12:13], v[20:21]\n v_fma_f64 v[12:13], v[12:13], v[14:15], v[22:23]\n v_fma_f64 v[14:15], v[14:15], v[16:17], v[24:25]\n v_fma_f64 v[16:17], v[16:17], v[10:11], v[26:27]")
KERNEL(mul_vv,     "v_mul_f64 v[30:31], v[10:11], v[20:21]\n v_mul_f64 v[32:33], v[12:13], v[22:23]\n v_mul_f64 v[34:35], v[14:15], v[24:25]\n v_mul_f64 v[36:37], v[16:17], v[26:27]")
KERNEL(mul_vs,     "v_mul_f64 v[30:31], v[10:11], s[20:21]\n v_mul_f64 v[32:33], v[12:13], s[20:21]\n v_mul_f64 v[34:35], v[14:15], s[20:21]\n v_mul_f64 v[36:37], v[16:17], s[20:21]")
KERNEL(add_vv,     "v_add_f64 v[30:31], v[10:11], v[20:21]\n v_add_f64 v[32:33], v[12:13], v[22:23]\n v_add_f64 v[34:35], v[14:15], v[24:25]\n v_add_f64 v[36:37], v[16:17], v[26:27]")
KERNEL(max_vv,     "v_max_f64 v[30:31], v[10:11], v[20:21]\n v_max_f64 v[32:33], v[12:13], v[22:23]\n v_max_f64 v[34:35], v[14:15], v[24:25]\n v_max_f64 v[36:37], v[16:17], v[26:27]")
KERNEL(fma32_vvv,  "v_fma_f32 v10, v10, v20, v30\n v_fma_f32 v12, v12, v22, v32\n v_fma_f32 v14, v14, v24, v34\n v_fma_f32 v16, v16, v26, v36")
KERNEL(pkfma32,    "v_pk_fma_f32 v[10:11], v[10:11], v[20:21], v[30:31]\n v_pk_fma_f32 v[12:13], v[12:13], v[22:23], v[32:33]\n v_pk_fma_f32 v[14:15], v[14:15], v[24:25], v[34:35]\n v_pk_fma_f32 v[16:17], v[16:17], v[26:27], v[36:37]")
KERNEL(mov64,      "v_mov_b64 v[30:31], v[10:11]\n v_mov_b64 v[32:33], v[12:13]\n v_mov_b64 v[34:35], v[14:15]\n v_mov_b64 v[36:37], v[16:17]")
KERNEL(cndmask,    "v_cndmask_b32 v30, v10, v20, vcc\n v_cndmask_b32 v32, v12, v22, vcc\n v_cndmask_b32 v34, v14, v24, vcc\n v_cndmask_b32 v36, v16, v26, vcc")

template <typename K>
void run(const char* name, K kern) {
  const int iters = 4000, blocks = 1024;
  double* out; unsigned long long* ticks;
  (void)hipMalloc(&out, blocks * 64 * sizeof(double));
  (void)hipMalloc(&ticks, blocks * sizeof(unsigned long long));
  kern<<<blocks, 64>>>(out, ticks, iters);
  kern<<<blocks, 64>>>(out, ticks, iters);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  printf("%-14s %6.2f ticks per instruction\n", name, avg / (iters * 16.0));
  (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
  run("fma_v_s_s", fma_v_s_s); run("fma_v_v_s", fma_v_v_s); run("fma_v_v_v", fma_v_v_v); run("fmac_vv", fmac_vv);
  run("fma_banks_a", fma_banks_a); run("fma_banks_b", fma_banks_b);
  run("mul_vv", mul_vv); run("mul_vs", mul_vs); run("add_vv", add_vv); run("max_vv", max_vv);
  run("fma32_vvv", fma32_vvv); run("pkfma32", pkfma32); run("mov64", mov64); run("cndmask", cndmask);
  return 0;
}
