// VALU issue-rate probe for gfx950: cycles per wave64 instruction per SIMD for the integer ops the k-mer hash is made of.
// build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate ; run on the GPU box.  Test infrastructure only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(X) X X X X X X X X
#define ITER 32768

#define KERNEL(NAME, ASM, CONSTR64)                                                                   \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                          \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    uint32_t c = seed | 1u;                                                                             \
    for (int i = 0; i < ITER; i++) {                                                                    \
      asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); \
    }                                                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                \
  }

#define A_ADD(R) "v_add_u32 " #R ", " #R ", %8\n"
#define A_XOR(R) "v_xor_b32 " #R ", " #R ", %8\n"
#define A_MULLO(R) "v_mul_lo_u32 " #R ", " #R ", %8\n"
#define A_MULHI(R) "v_mul_hi_u32 " #R ", " #R ", %8\n"
#define A_MUL24(R) "v_mul_u32_u24 " #R ", " #R ", %8\n"
#define A_MAD24(R) "v_mad_u32_u24 " #R ", " #R ", %8, %8\n"
#define A_ALIGN(R) "v_alignbit_b32 " #R ", " #R ", %8, 7\n"
#define A_ADD3(R) "v_add3_u32 " #R ", " #R ", %8, %8\n"
#define A_PERM(R) "v_perm_b32 " #R ", " #R ", %8, %8\n"
#define A_LSHLADD(R) "v_lshl_add_u32 " #R ", " #R ", 3, %8\n"
#define A_BFE(R) "v_bfe_u32 " #R ", " #R ", 3, 17\n"
#define A_CNDMASK(R) "v_cndmask_b32 " #R ", " #R ", %8, vcc\n"
#define A_MBCNT(R) "v_mbcnt_lo_u32_b32 " #R ", " #R ", %8\n"

KERNEL(k_add, A_ADD, 0)
KERNEL(k_xor, A_XOR, 0)
KERNEL(k_mullo, A_MULLO, 0)
KERNEL(k_mulhi, A_MULHI, 0)
KERNEL(k_mul24, A_MUL24, 0)
KERNEL(k_mad24, A_MAD24, 0)
KERNEL(k_align, A_ALIGN, 0)
KERNEL(k_add3, A_ADD3, 0)
KERNEL(k_perm, A_PERM, 0)
KERNEL(k_lshladd, A_LSHLADD, 0)
KERNEL(k_bfe, A_BFE, 0)
KERNEL(k_cndmask, A_CNDMASK, 0)
KERNEL(k_mbcnt, A_MBCNT, 0)
#define A_CNDS(R) "v_cndmask_b32_e64 " #R ", " #R ", %8, s[10:11]\n"
#define A_LSHR(R) "v_lshrrev_b32 " #R ", 1, " #R "\n"
#define A_MIX(R) "v_mul_lo_u32 " #R ", " #R ", %8\nv_xor_b32 " #R ", " #R ", %8\n"
#define A_MIX2(R) "v_mad_u64_u32 v[20:21], vcc, " #R ", %8, 0\nv_add_u32 " #R ", " #R ", v20\n"
#define A_MOV(R) "v_mov_b32 " #R ", %8\n"
KERNEL(k_cnds, A_CNDS, 0)
KERNEL(k_lshr, A_LSHR, 0)
KERNEL(k_mix, A_MIX, 0)
KERNEL(k_mov, A_MOV, 0)

// 64-bit destinations
#define KERNEL64(NAME, ASM)                                                                            \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                          \
    uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    uint32_t c = seed | 1u; uint64_t c64 = ((uint64_t)c << 32) | c;                                     \
    for (int i = 0; i < ITER; i++) {                                                                    \
      asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c64) : "vcc", "v20", "v21", "s10", "s11"); \
    }                                                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);    \
  }
#define A_MAD64(R) "v_mad_u64_u32 " #R ", vcc, %8, %8, " #R "\n"
#define A_LSHLADD64(R) "v_lshl_add_u64 " #R ", " #R ", 0, %9\n"
#define A_LSHL64(R) "v_lshlrev_b64 " #R ", 3, " #R "\n"
#define A_CMP64(R) "v_cmp_lt_u64 vcc, " #R ", %9\n"
#define A_MOV64(R) "v_mov_b64 " #R ", %9\n"
KERNEL64(k_mad64, A_MAD64)
KERNEL64(k_lshladd64, A_LSHLADD64)
KERNEL64(k_lshl64, A_LSHL64)
KERNEL64(k_cmp64, A_CMP64)
// canonical-min idiom: compare + two selects, through VCC (what hipcc emits) and through an SGPR pair
#define A_MINV(R) "v_cmp_lt_u64 vcc, " #R ", %9\nv_cndmask_b32 v20, v20, %8, vcc\nv_cndmask_b32 v21, v21, %8, vcc\n"
#define A_MINS(R) "v_cmp_lt_u64 s[10:11], " #R ", %9\nv_cndmask_b32_e64 v20, v20, %8, s[10:11]\nv_cndmask_b32_e64 v21, v21, %8, s[10:11]\n"
#define A_MINV64(R) "v_cmp_lt_u64_e64 vcc, " #R ", %9\nv_cndmask_b32_e64 v20, v20, %8, vcc\nv_cndmask_b32_e64 v21, v21, %8, vcc\n"
#define A_ADDC(R) "v_cmp_lt_u64 vcc, " #R ", %9\nv_addc_co_u32 v20, vcc, v20, %8, vcc\n"
KERNEL64(k_minv, A_MINV)
KERNEL64(k_mins, A_MINS)
KERNEL64(k_minv64, A_MINV64)
KERNEL64(k_addc, A_ADDC)

typedef void (*kern_t)(uint32_t*, uint32_t);
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
  printf("device %s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
  uint32_t* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
  for (int wavesPerSimd : {4, 8}) {
  const int blocks = cus * wavesPerSimd;      // 256 threads = 4 waves = one per SIMD
  printf("---- %d waves per SIMD\n", wavesPerSimd);
  
  struct { const char* name; kern_t k; } ks[] = {
    {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24},
    {"v_mad_u32_u24", k_mad24}, {"v_alignbit_b32", k_align}, {"v_add3_u32", k_add3}, {"v_perm_b32", k_perm}, {"v_lshl_add_u32", k_lshladd},
    {"v_bfe_u32", k_bfe}, {"v_cndmask_b32", k_cndmask}, {"v_mbcnt_lo", k_mbcnt}, {"v_cndmask(sgpr)", k_cnds}, {"v_lshrrev_b32", k_lshr}, {"mul_lo+xor (2 instr)", k_mix}, {"v_mov_b32", k_mov}, {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshladd64},
    {"v_lshlrev_b64", k_lshl64}, {"v_cmp_lt_u64", k_cmp64}, {"cmp+2cndmask vcc e32 (3)", k_minv}, {"cmp+2cndmask sgpr (3)", k_mins}, {"cmp+2cndmask vcc e64 (3)", k_minv64}, {"cmp+addc vcc (2)", k_addc}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrPerSimd = (double)wavesPerSimd * ITER * 8;
    printf("%-16s %8.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", k.name, ms, ms * 1e-3 * ghz * 1e9 / instrPerSimd, ghz);
  }
  }
  return 0;
}
