// VALU / SALU issue-rate microbenchmark for gfx950 (MI355X): cycles per wave64 instruction per SIMD for the integer, permute,
// funnel-shift and DPP instructions the LINEMOD kernels are made of, at 1, 2, 4 and 8 resident waves per SIMD.
// Settles the "2 or 4 cycles per wave64 VALU instruction" question behind the roofline of k_score_coarse_u8 /
// k_color_quantize (/opt/skills/guides/MI355X_MICROARCH.md "Wave scheduling" and "vector-instruction
// ISSUE cost").  Every wave runs ITERS x 64 instructions, 8 independent dependency chains, timed with s_memtime (shader clock).
//   build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip      run: ./valu_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 2000;

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

// one asm statement per instruction keeps the register allocation with the compiler; "+v" chains are independent of each other
#define OP_ADD(k)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_AND(k)      asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_LSHR(k)     asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[k]));
#define OP_PERM(k)     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_ALIGNBIT(k) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_ALIGNBIT_S(k) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(sc));
#define OP_DPP(k)      asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(b));
#define OP_MUL24(k)    asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_MAD24(k)    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_DOT2(k)     asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_PKMUL(k)    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_ADD3(k)     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_ANDOR(k)    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_LSHLADD(k)  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
#define OP_BFE(k)      asm volatile("v_bfe_u32 %0, %0, 4, 8" : "+v"(a[k]));
#define OP_MULLO(k)    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_CVT(k)      asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k]));
#define OP_FMA(k)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_BITOP3(k)   asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_BITOP3_S(k) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf8" : "+v"(a[k]) : "s"(sc), "v"(c));
#define OP_BITOP3_SL(k) asm volatile("v_bitop3_b32 %0, %0, %1, 7 bitop3:0x80" : "+v"(a[k]) : "s"(sc));
#define OP_ANDOR_S(k)  asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(sc), "v"(c));
#define OP_ADD3_L(k)   asm volatile("v_add3_u32 %0, %0, %1, 1" : "+v"(a[k]) : "v"(b));
#define OP_AND_S(k)    asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[k]) : "s"(sc));
#define OP_LSHR_S(k)   asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[k]) : "s"(sc));
#define OP_ADD_L(k)    asm volatile("v_add_u32 %0, 0x7f7f7f7f, %0" : "+v"(a[k]));
#define OP_BCNT(k)     asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_DOT4(k)     asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_DOT4_S(k)   asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[k]) : "s"(sc), "v"(c));
#define OP_SDOT2(k)    asm volatile("v_dot2_i32_i16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_PKMAD(k)    asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_PKADD(k)    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_PACK(k)     asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_CNDMASK(k)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
#define OP_MAXI(k)     asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_FFBL(k)     asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[k]));
#define OP_XOR(k)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_CMP(k)      asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define OP_MUL24_SDWA(k) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a[k]) : "v"(b));
#define OP_LSHL_OR(k)  asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[k]) : "v"(b));
// select idioms (round 4: v_cndmask_b32 reading a stale VCC measured 22.8 cycles per instruction; what does a real compare + select cost?)
#define OP_CMP_CND(k)  asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
#define OP_CMP_NOP_CND(k) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
#define OP_CMP64_CND(k) asm volatile("v_cmp_gt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(a[k]) : "v"(b), "v"(c) : "s20", "s21");
#define OP_CND_S(k)    asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[k]) : "v"(b) : );
#define OP_SEL_ARITH(k) asm volatile("v_sub_u32 %1, %2, %0\n v_ashrrev_i32 %1, 31, %1\n v_bitop3_b32 %0, %0, %3, %1 bitop3:0xe4" : "+v"(a[k]), "+v"(d[k]) : "v"(b), "v"(c));
#define OP_CMP2_CND2(k) asm volatile("v_cmp_gt_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(a[k]), "+v"(d[k]) : "v"(b), "v"(c) : "vcc");
#define OP_MAXU(k)     asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_SUB(k)      asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_ASHR(k)     asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[k]));
#define OP_LSHL(k)     asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_ADDC(k)     asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[k]) : : "vcc");
#define OP_READLANE(k) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s[k]) : "v"(b));
#define OP_SAND(k)     asm volatile("s_and_b32 %0, %0, %1" : "+s"(s[k]) : "s"(sc) : "scc");
#define OP_SLSHR(k)    asm volatile("s_lshr_b32 %0, %0, 1" : "+s"(s[k]) : : "scc");
// what k_score_coarse_u8 does per table entry: v_readlane + two scalar ops
#define OP_RL_SALU(k)  asm volatile("v_readlane_b32 %0, %1, 5\n s_lshr_b32 %2, %0, 27\n s_and_b32 %0, %0, 0x7ffffff" : "=s"(s[k]), "+v"(b), "=s"(t[k]) : : "scc");
// the inner step of k_score_coarse_u8 per group of 3 loaded dwords (FAST group): 2 adds, dpp, alignbit, and, lshr, and, 2 adds = 9
// VALU; the two s_nop stand for the wait states the DPP read of a freshly written VGPR needs (the compiler fills them with
// independent work in the real kernel)
#define OP_SCORE(k)    asm volatile("v_add_u32 %1, %3, %4\n v_add_u32 %1, %1, %0\n s_nop 1\n v_mov_b32_dpp %2, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n" \
                                    "v_alignbit_b32 %2, %2, %1, %5\n v_and_b32 %1, 0x0f0f0f0f, %2\n v_lshrrev_b32 %2, 4, %2\n v_and_b32 %2, 0x0f0f0f0f, %2\n" \
                                    "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %2\n" : "+v"(a[k]), "+v"(d[k]), "+v"(e[k]) : "v"(b), "v"(c), "s"(sc));

template <int OP>
__global__ void k_issue(unsigned long long* out, uint32_t* sink, uint32_t sc_in) {
  uint32_t a[8], d[8], e[8], s[8], t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t b = threadIdx.x * 2654435761u + 12345u, c = threadIdx.x ^ 0x01020304u;
  const uint32_t sc = __builtin_amdgcn_readfirstlane(sc_in);
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x + k * 977u; d[k] = k; e[k] = k; s[k] = sc_in + k; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if (OP == 0) { BODY8(OP_ADD) }
    if (OP == 1) { BODY8(OP_AND) }
    if (OP == 2) { BODY8(OP_LSHR) }
    if (OP == 3) { BODY8(OP_PERM) }
    if (OP == 4) { BODY8(OP_ALIGNBIT) }
    if (OP == 5) { BODY8(OP_ALIGNBIT_S) }
    if (OP == 6) { BODY8(OP_DPP) }
    if (OP == 7) { BODY8(OP_MUL24) }
    if (OP == 8) { BODY8(OP_MAD24) }
    if (OP == 9) { BODY8(OP_DOT2) }
    if (OP == 10) { BODY8(OP_PKMUL) }
    if (OP == 11) { BODY8(OP_ADD3) }
    if (OP == 12) { BODY8(OP_ANDOR) }
    if (OP == 13) { BODY8(OP_LSHLADD) }
    if (OP == 14) { BODY8(OP_BFE) }
    if (OP == 15) { BODY8(OP_MULLO) }
    if (OP == 16) { BODY8(OP_CVT) }
    if (OP == 17) { BODY8(OP_FMA) }
    if (OP == 18) { BODY8(OP_READLANE) }
    if (OP == 19) { BODY8(OP_SAND) }
    if (OP == 20) { BODY8(OP_SLSHR) }
    if (OP == 21) { R8(OP_SCORE) }   // 8 x 9 = 72 VALU instructions
    if (OP == 22) { BODY8(OP_RL_SALU) }  // 64 x (1 VALU + 2 SALU)
    if (OP == 23) { BODY8(OP_BITOP3) }
    if (OP == 24) { BODY8(OP_BITOP3_S) }
    if (OP == 25) { BODY8(OP_BITOP3_SL) }
    if (OP == 26) { BODY8(OP_ANDOR_S) }
    if (OP == 27) { BODY8(OP_ADD3_L) }
    if (OP == 28) { BODY8(OP_AND_S) }
    if (OP == 29) { BODY8(OP_LSHR_S) }
    if (OP == 30) { BODY8(OP_ADD_L) }
    if (OP == 31) { BODY8(OP_BCNT) }
    if (OP == 32) { BODY8(OP_DOT4) }
    if (OP == 33) { BODY8(OP_DOT4_S) }
    if (OP == 34) { BODY8(OP_SDOT2) }
    if (OP == 35) { BODY8(OP_PKMAD) }
    if (OP == 36) { BODY8(OP_PKADD) }
    if (OP == 37) { BODY8(OP_PACK) }
    if (OP == 38) { BODY8(OP_CNDMASK) }
    if (OP == 39) { BODY8(OP_MAXI) }
    if (OP == 40) { BODY8(OP_FFBL) }
    if (OP == 41) { BODY8(OP_XOR) }
    if (OP == 42) { BODY8(OP_CMP) }
    if (OP == 43) { BODY8(OP_MUL24_SDWA) }
    if (OP == 44) { BODY8(OP_LSHL_OR) }
    if (OP == 45) { BODY8(OP_CMP_CND) }
    if (OP == 46) { BODY8(OP_CMP_NOP_CND) }
    if (OP == 47) { BODY8(OP_CMP64_CND) }
    if (OP == 48) { BODY8(OP_CND_S) }
    if (OP == 49) { BODY8(OP_SEL_ARITH) }
    if (OP == 50) { BODY8(OP_CMP2_CND2) }
    if (OP == 51) { BODY8(OP_MAXU) }
    if (OP == 52) { BODY8(OP_SUB) }
    if (OP == 53) { BODY8(OP_ASHR) }
    if (OP == 54) { BODY8(OP_LSHL) }
    if (OP == 55) { BODY8(OP_ADDC) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t acc = b + c;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc += a[k] + d[k] + e[k] + s[k] + t[k];
  if (acc == 0x12345678u) sink[0] = acc;  // keeps every chain alive
  if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

struct Op { const char* name; void (*fn)(unsigned long long*, uint32_t*, uint32_t); int per_iter = 64; };

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# device %s, %d CUs, clockRate %d kHz; ITERS %d x 64 instructions per wave, 8 independent chains\n", prop.gcnArchName, cus, prop.clockRate, ITERS);
  printf("# cycles = median over waves of s_memtime delta; cyc/instr/SIMD = cycles / (ITERS*64*waves_per_SIMD)\n");
  unsigned long long* d_out = nullptr;
  uint32_t* d_sink = nullptr;
  const size_t max_waves = (size_t)cus * 32;
  CHECK(hipMalloc(&d_out, max_waves * 8));
  CHECK(hipMalloc(&d_sink, 64));
  const Op ops[] = {
      {"v_add_u32", k_issue<0>},         {"v_and_b32", k_issue<1>},        {"v_lshrrev_b32", k_issue<2>},   {"v_perm_b32", k_issue<3>},
      {"v_alignbit_b32 (vgpr sh)", k_issue<4>}, {"v_alignbit_b32 (sgpr sh)", k_issue<5>}, {"v_mov_b32_dpp wave_shl", k_issue<6>}, {"v_mul_u32_u24", k_issue<7>},
      {"v_mad_u32_u24", k_issue<8>},     {"v_dot2_u32_u16", k_issue<9>},   {"v_pk_mul_lo_u16", k_issue<10>}, {"v_add3_u32", k_issue<11>},
      {"v_and_or_b32", k_issue<12>},     {"v_lshl_add_u32", k_issue<13>},  {"v_bfe_u32", k_issue<14>},      {"v_mul_lo_u32", k_issue<15>},
      {"v_cvt_f32_i32", k_issue<16>},    {"v_fma_f32", k_issue<17>},       {"v_readlane_b32", k_issue<18>}, {"s_and_b32", k_issue<19>},
      {"s_lshr_b32", k_issue<20>},       {"score step: 9 VALU per 3 dwords", k_issue<21>, 72}, {"v_readlane + 2 SALU (per triple)", k_issue<22>},
      {"v_bitop3_b32 (v,v,v)", k_issue<23>}, {"v_bitop3_b32 (v,s,v)", k_issue<24>}, {"v_bitop3_b32 (v,s,inline const)", k_issue<25>}, {"v_and_or_b32 (v,s,v)", k_issue<26>},
      {"v_add3_u32 (v,v,inline const)", k_issue<27>}, {"v_and_b32 (s,v)", k_issue<28>}, {"v_lshrrev_b32 (s,v)", k_issue<29>}, {"v_add_u32 (literal,v)", k_issue<30>},
      {"v_bcnt_u32_b32", k_issue<31>},
      // round 4: the instructions the rewritten colour quantiser is made of
      {"v_dot4_u32_u8 (v,v,v)", k_issue<32>}, {"v_dot4_u32_u8 (v,s,v)", k_issue<33>}, {"v_dot2_i32_i16", k_issue<34>}, {"v_pk_mad_u16", k_issue<35>},
      {"v_pk_add_u16", k_issue<36>}, {"v_pack_b32_f16", k_issue<37>}, {"v_cndmask_b32 (vcc)", k_issue<38>}, {"v_max_i32", k_issue<39>},
      {"v_ffbl_b32", k_issue<40>}, {"v_xor_b32", k_issue<41>}, {"v_cmp_gt_u32 (vcc)", k_issue<42>}, {"v_mul_u32_u24_sdwa", k_issue<43>},
      {"v_lshl_or_b32", k_issue<44>},
      {"v_cmp vcc + v_cndmask vcc (2 instr)", k_issue<45>, 128}, {"v_cmp + s_nop 1 + v_cndmask (2 instr)", k_issue<46>, 128},
      {"v_cmp sgpr pair + v_cndmask e64 (2)", k_issue<47>, 128}, {"v_cndmask_b32 (stale sgpr pair)", k_issue<48>},
      {"sub + ashr + bitop3 select (3)", k_issue<49>, 192}, {"v_cmp + 2 x v_cndmask (3 instr)", k_issue<50>, 192},
      {"v_max_u32", k_issue<51>}, {"v_sub_u32", k_issue<52>}, {"v_ashrrev_i32", k_issue<53>}, {"v_lshlrev_b32 (v,v)", k_issue<54>}, {"v_addc_co_u32 (vcc in/out)", k_issue<55>}};
  printf("# per cell: A / B.  A = median over waves of the s_memtime delta / (instructions per wave x waves per SIMD);\n");
  printf("#           B = from wall time: kernel time (HIP events) x shader clock / (instructions per SIMD), shader clock = the\n");
  printf("#               longest wave's s_memtime delta / kernel time of the same launch (robust to uneven workgroup placement)\n");
  printf("%-34s %13s %13s %13s %13s   (cycles per wave64 instruction per SIMD at 1/2/4/8 waves per SIMD)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double clock_sum = 0; int clock_n = 0;
  for (const Op& op : ops) {
    printf("%-34s", op.name);
    for (int wps : {1, 2, 4, 8}) {
      // wps <= 4: one workgroup of 4*wps waves per CU; 8: two workgroups of 16 waves per CU
      const int threads = 64 * 4 * (wps == 8 ? 4 : wps);
      const int blocks = cus * (wps == 8 ? 2 : 1);
      hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, 4u);  // warms the instruction cache
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, 4u);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const size_t n = (size_t)blocks * threads / 64;
      std::vector<unsigned long long> h(n);
      CHECK(hipMemcpy(h.data(), d_out, n * 8, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double med = (double)h[n / 2], longest = (double)h[n - 1];
      const double instr_per_simd = (double)ITERS * op.per_iter * wps;
      const double ghz = longest / (ms * 1e6);
      clock_sum += ghz; ++clock_n;
      printf("  %5.2f / %5.2f", med / instr_per_simd, (ms * 1e-3) * (ghz * 1e9) / instr_per_simd);
    }
    printf("\n");
    fflush(stdout);
  }
  printf("# mean shader clock over all launches (longest wave's s_memtime delta / kernel time): %.2f GHz (a lower bound: launch overhead is in the kernel time)\n", clock_sum / clock_n);
  return 0;
}
