// Instruction-rate micro-benchmarks for gfx950 (MI355X): decides which pipe bounds the RTCSM3D
// score kernel.  Every kernel runs 256-thread workgroups (one wave per SIMD
// per workgroup), WPS workgroups per CU; the loop body is UNROLL independent instructions issued
// back to back.  Output: cycles per wave-instruction per SIMD from s_memtime (shader clock) and
// from the wall clock at the measured effective frequency.
//   hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int kIters = 2000;

__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// 8 independent register chains, 32 instructions per iteration
#define VALU_KERNEL(NAME, ASM)                                                                       \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* cyc, float* sink) {               \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,       \
          a6 = a0 + 6, a7 = a0 + 7;                                                                   \
    float b = 1.0001f, c = 0.5f;                                                                      \
    const unsigned long long t0 = memtime();                                                         \
    for (int i = 0; i < kIters; ++i) {                                                               \
      asm volatile(REP8(ASM(0) ASM(1) ASM(2) ASM(3))                                                 \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                   : "v"(b), "v"(c));                                                                \
      asm volatile(REP8(ASM(4) ASM(5) ASM(6) ASM(7))                                                 \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                   : "v"(b), "v"(c));                                                                \
    }                                                                                                \
    const unsigned long long t1 = memtime();                                                         \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) sink[0] = a0;                          \
  }
// note: two asm blocks x 32 = 64 instructions per iteration

#define A_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define A_ADD(k) "v_add_f32 %" #k ", %" #k ", %9\n"
#define A_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define A_CVTFLR(k) "v_cvt_flr_i32_f32 %" #k ", %" #k "\n"
#define A_FRACT(k) "v_fract_f32 %" #k ", %" #k "\n"
#define A_FLOOR(k) "v_floor_f32 %" #k ", %" #k "\n"
#define A_MIN3(k) "v_min3_f32 %" #k ", %" #k ", %8, %9\n"
#define A_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %9\n"
#define A_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define A_ADDU(k) "v_add_u32 %" #k ", %" #k ", %9\n"
#define A_CVTI(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
#define A_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define A_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define A_CMPCND(k) "v_cmp_ge_f32 vcc, %" #k ", %9\n v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"

#define A_MINU(k) "v_min_u32 %" #k ", %" #k ", %9\n"
#define A_MINU_SDWA(k) "v_min_u32_sdwa %" #k ", %" #k ", %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
#define A_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %9\n"
#define A_MUL24_SDWA(k) "v_mul_u32_u24_sdwa %" #k ", %" #k ", %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define A_ADDU_SDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define A_LSHL(k) "v_lshlrev_b32 %" #k ", 3, %" #k "\n"
#define A_LSHR(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define A_AND(k) "v_and_b32 %" #k ", %" #k ", %9\n"
#define A_MAXF(k) "v_max_f32 %" #k ", %" #k ", %9\n"
#define A_SUBF(k) "v_sub_f32 %" #k ", %" #k ", %9\n"
#define A_FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define A_CMP(k) "v_cmp_le_u32 vcc, %" #k ", %9\n"
#define A_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 7\n"
#define A_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define A_MADU16(k) "v_mad_u32_u16 %" #k ", %" #k ", %8, %9 op_sel:[1,0,0,0]\n"
#define A_DOT2(k) "v_dot2_u32_u16 %" #k ", %8, %9, %" #k "\n"
#define A_PKADDU16(k) "v_pk_add_u16 %" #k ", %" #k ", %9\n"
#define A_PKMINU16(k) "v_pk_min_u16 %" #k ", %" #k ", %9\n"
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %9\n"
#define A_MULDEN(k) "v_mul_f32 %" #k ", 0x00000003, %" #k "\n"
#define A_MOV(k) "v_mov_b32 %" #k ", %9\n"
#define A_CVTU(k) "v_cvt_u32_f32 %" #k ", %" #k "\n"
#define A_RNDNE(k) "v_rndne_f32 %" #k ", %" #k "\n"
#define A_XOR(k) "v_xor_b32 %" #k ", %" #k ", %9\n"
#define A_SUBU(k) "v_sub_u32 %" #k ", %" #k ", %9\n"
#define A_MAX3U(k) "v_max3_u32 %" #k ", %" #k ", %8, %9\n"
#define A_ADDLSHL(k) "v_add_lshl_u32 %" #k ", %" #k ", %9, 1\n"
VALU_KERNEL(k_minu, A_MINU)
VALU_KERNEL(k_minu_sdwa, A_MINU_SDWA)
VALU_KERNEL(k_mul24, A_MUL24)
VALU_KERNEL(k_mul24_sdwa, A_MUL24_SDWA)
VALU_KERNEL(k_addu_sdwa, A_ADDU_SDWA)
VALU_KERNEL(k_lshl, A_LSHL)
VALU_KERNEL(k_lshr, A_LSHR)
VALU_KERNEL(k_and, A_AND)
VALU_KERNEL(k_maxf, A_MAXF)
VALU_KERNEL(k_subf, A_SUBF)
VALU_KERNEL(k_fmac, A_FMAC)
VALU_KERNEL(k_cmp, A_CMP)
VALU_KERNEL(k_bfe, A_BFE)
VALU_KERNEL(k_andor, A_ANDOR)
VALU_KERNEL(k_madu16, A_MADU16)
VALU_KERNEL(k_dot2, A_DOT2)
VALU_KERNEL(k_pkaddu16, A_PKADDU16)
VALU_KERNEL(k_pkminu16, A_PKMINU16)
VALU_KERNEL(k_mullo, A_MULLO)
VALU_KERNEL(k_mulden, A_MULDEN)
VALU_KERNEL(k_mov, A_MOV)
VALU_KERNEL(k_cvtu, A_CVTU)
VALU_KERNEL(k_rndne, A_RNDNE)
VALU_KERNEL(k_xor, A_XOR)
VALU_KERNEL(k_subu, A_SUBU)
VALU_KERNEL(k_max3u, A_MAX3U)
VALU_KERNEL(k_addlshl, A_ADDLSHL)
VALU_KERNEL(k_fma, A_FMA)
VALU_KERNEL(k_add, A_ADD)
VALU_KERNEL(k_mul, A_MUL)
VALU_KERNEL(k_add3, A_ADD3)
VALU_KERNEL(k_cvtflr, A_CVTFLR)
VALU_KERNEL(k_cvti, A_CVTI)
VALU_KERNEL(k_fract, A_FRACT)
VALU_KERNEL(k_floor, A_FLOOR)
VALU_KERNEL(k_min3, A_MIN3)
VALU_KERNEL(k_med3, A_MED3)
VALU_KERNEL(k_lshladd, A_LSHLADD)
VALU_KERNEL(k_mad24, A_MAD24)
VALU_KERNEL(k_addu, A_ADDU)
VALU_KERNEL(k_cmpcnd, A_CMPCND)

__global__ __launch_bounds__(256) void k_add_sgpr(unsigned long long* cyc, float* sink, float sv) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned long long t0 = memtime();
  for (int i = 0; i < kIters; ++i) {
#define AS(k) "v_add_f32 %" #k ", %8, %" #k "\n"
    asm volatile(REP8(AS(0) AS(1) AS(2) AS(3))
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "s"(sv));
    asm volatile(REP8(AS(4) AS(5) AS(6) AS(7))
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "s"(sv));
  }
  const unsigned long long t1 = memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) sink[0] = a0;
}

// u16 gathers out of an LDS box with the score kernel's lane pattern: lane = candidate rotation
// (i = lane % 11, j = lane / 11), cell offset (floor(i s), floor(j s), floor((i + j) s / 3)) for a spread of
// s cells per rotation step; box strides (sx, sxy) in cells.
__global__ __launch_bounds__(256) void k_lds_box(unsigned long long* cyc, float* sink, float spread, int sx, int sxy) {
  __shared__ unsigned lds[12288];
  for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int i = lane % 11, j = lane / 11;
  const int cx = static_cast<int>(i * spread), cy = static_cast<int>(j * spread), cz = static_cast<int>((i + j) * spread / 3.f);
  const unsigned addr = (((2 + cz) * sxy) + (2 + cy) * sx + (2 + cx)) * 2u;
  unsigned acc = 0;
  const unsigned long long t0 = memtime();
  for (int it = 0; it < kIters; ++it) {
    unsigned v0, v1, v2, v3, v4, v5, v6, v7;
#define LD(k, off) "ds_read_u16 %" #k ", %8 offset:" #off "\n"
    asm volatile(LD(0, 0) LD(1, 2) LD(2, 6) LD(3, 10) LD(4, 120) LD(5, 250) LD(6, 1300) LD(7, 2602) "s_waitcnt lgkmcnt(0)\n"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                 : "v"(addr)
                 : "memory");
#undef LD
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  const unsigned long long t1 = memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

// uniform (scalar) loads of 16 bytes, 8 per iteration, from a 2 KB table
__global__ __launch_bounds__(256) void k_sload(unsigned long long* cyc, float* sink, const float4* table) {
  float acc = 0.f;
  const unsigned long long t0 = memtime();
  for (int it = 0; it < kIters; ++it) {
    const float4* t = table + (it & 15) * 8;
    float4 a0, a1, a2, a3, a4, a5, a6, a7;
    asm volatile("s_load_dwordx4 %0, %8, 0x0\n s_load_dwordx4 %1, %8, 0x10\n s_load_dwordx4 %2, %8, 0x20\n"
                 "s_load_dwordx4 %3, %8, 0x30\n s_load_dwordx4 %4, %8, 0x40\n s_load_dwordx4 %5, %8, 0x50\n"
                 "s_load_dwordx4 %6, %8, 0x60\n s_load_dwordx4 %7, %8, 0x70\n s_waitcnt lgkmcnt(0)\n"
                 : "=s"(a0), "=s"(a1), "=s"(a2), "=s"(a3), "=s"(a4), "=s"(a5), "=s"(a6), "=s"(a7)
                 : "s"(t)
                 : "memory");
    acc += a0.x + a1.y + a2.z + a3.w + a4.x + a5.y + a6.z + a7.w;
  }
  const unsigned long long t1 = memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 12345.678f) sink[0] = acc;
}

// packed fp32: 2 floats per lane per instruction, 64-bit register pairs
__global__ __launch_bounds__(256) void k_pkfma(unsigned long long* cyc, float* sink) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
     a6 = a0 + 6.f, a7 = a0 + 7.f;
  f2 b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
  const unsigned long long t0 = memtime();
  for (int i = 0; i < kIters; ++i) {
#define PK(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
    asm volatile(REP8(PK(0) PK(1) PK(2) PK(3))
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b), "v"(c));
    asm volatile(REP8(PK(4) PK(5) PK(6) PK(7))
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b), "v"(c));
  }
  const unsigned long long t1 = memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s.x + s.y == 12345.678f) sink[0] = s.x;
}

// LDS reads: 64 per iteration, addresses from `addr_mode`
//   0: every lane the same address (broadcast)   1: lane * 4 (conflict free)
//   2: u16 gather in a 24^3 box, lanes spread over a 5x5x5 neighbourhood
//   3: same, 3x3x3 neighbourhood                   4: 9x9x9 neighbourhood
template <int WIDTH>  // 32: ds_read_b32, 16: ds_read_u16
__global__ __launch_bounds__(256) void k_lds(unsigned long long* cyc, float* sink, int addr_mode) {
  __shared__ unsigned lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  unsigned h = lane * 2654435761u + 12345u;
  unsigned addr;
  auto box = [&](unsigned w) {
    const unsigned dx = (h >> 4) % w, dy = (h >> 12) % w, dz = (h >> 20) % w;
    return (((4 + dz) * 24 + (4 + dy)) * 24 + (4 + dx)) * 2u;
  };
  switch (addr_mode) {
    case 0: addr = 64; break;
    case 1: addr = lane * 4; break;
    case 2: addr = box(5); break;
    case 3: addr = box(3); break;
    default: addr = box(9); break;
  }
  unsigned acc = 0;
  const unsigned long long t0 = memtime();
  for (int i = 0; i < kIters; ++i) {
    unsigned v0, v1, v2, v3, v4, v5, v6, v7;
    if (WIDTH == 32) {
#define LD(k, off) "ds_read_b32 %" #k ", %8 offset:" #off "\n"
      asm volatile(LD(0, 0) LD(1, 4) LD(2, 8) LD(3, 12) LD(4, 16) LD(5, 20) LD(6, 24) LD(7, 28) "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(addr)
                   : "memory");
#undef LD
    } else {
#define LD(k, off) "ds_read_u16 %" #k ", %8 offset:" #off "\n"
      asm volatile(LD(0, 0) LD(1, 2) LD(2, 48) LD(3, 50) LD(4, 1152) LD(5, 1154) LD(6, 1200) LD(7, 1202) "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(addr)
                   : "memory");
#undef LD
    }
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  const unsigned long long t1 = memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

// global gathers of one u16 / u32 per lane out of an L1/L2-resident region
//   mode 0: all lanes one address; 1: lanes spread inside ONE 128-B line; 2: over 4 lines;
//   3: over 16 lines; 4: every lane its own line; 5: coalesced (lane * WIDTH/8)
template <int WIDTH>
__global__ __launch_bounds__(256) void k_gather(unsigned long long* cyc, float* sink, const char* base, int mode) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned h = lane * 2654435761u + 777u;
  unsigned off;
  const unsigned within = ((h >> 8) % 32) * 4;  // 4-byte aligned slot inside a line
  switch (mode) {
    case 0: off = 256; break;
    case 1: off = within; break;
    case 2: off = ((h >> 20) % 4) * 128 + within; break;
    case 3: off = ((h >> 20) % 16) * 128 + within; break;
    case 4: off = lane * 128 + within; break;
    default: off = lane * (WIDTH / 8); break;
  }
  off += (blockIdx.x % 64) * 16384;  // every block its own 16 KB window of the buffer
  const char* p = base + off;
  unsigned acc = 0;
  const unsigned long long t0 = memtime();
  for (int i = 0; i < kIters / 4; ++i) {
    unsigned v0, v1, v2, v3, v4, v5, v6, v7;
    if (WIDTH == 16) {
#define LD(k, o) "global_load_ushort %" #k ", %8, off offset:" #o "\n"
      asm volatile(LD(0, 0) LD(1, 512) LD(2, 1024) LD(3, 1536) LD(4, 2048) LD(5, 2560) LD(6, 3072) LD(7, 3584) "s_waitcnt vmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(p)
                   : "memory");
#undef LD
    } else {
#define LD(k, o) "global_load_dword %" #k ", %8, off offset:" #o "\n"
      asm volatile(LD(0, 0) LD(1, 512) LD(2, 1024) LD(3, 1536) LD(4, 2048) LD(5, 2560) LD(6, 3072) LD(7, 3584) "s_waitcnt vmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(p)
                   : "memory");
#undef LD
    }
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  const unsigned long long t1 = memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

struct Result {
  double wall_ms;
  double max_cyc;
};

template <typename F>
Result run(F launch, int blocks, unsigned long long* d_cyc) {
  std::vector<unsigned long long> h(blocks * 4);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();  // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return {ms, mx};
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n \"rows\": [\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  unsigned long long* d_cyc;
  float* d_sink;
  char* d_buf;
  CHECK(hipMalloc(&d_cyc, 8 * 4 * cus * 8));
  CHECK(hipMalloc(&d_sink, 64));
  CHECK(hipMalloc(&d_buf, 64 * 16384 + 8192));
  CHECK(hipMemset(d_buf, 1, 64 * 16384 + 8192));
  bool first = true;
  auto report = [&](const char* name, int wps, double instr_per_wave, Result r) {
    // cycles per wave-instruction per SIMD: wps waves share one SIMD
    const double cyc_per_instr = r.max_cyc / (instr_per_wave * wps);
    const double wall_cyc_24 = r.wall_ms * 1e-3 * 2.4e9 / (instr_per_wave * wps);
    printf("%s  {\"name\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_wave_instr_per_simd_memtime\": %.3f, "
           "\"same_at_2.4GHz_wall\": %.3f, \"wall_ms\": %.4f}",
           first ? "" : ",\n", name, wps, cyc_per_instr, wall_cyc_24, r.wall_ms);
    first = false;
  };
#define RUN_VALU(K, LABEL, PER_ITER)                                                                  \
  for (int wps : {1, 2, 4, 8}) {                                                                      \
    const int blocks = cus * wps;                                                                     \
    report(LABEL, wps, double(kIters) * PER_ITER,                                                     \
           run([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink); }, blocks, d_cyc)); \
  }
  const bool quick = getenv("UBENCH_FULL") == nullptr;  // round 2b: only the new rows unless UBENCH_FULL is set
  RUN_VALU(k_minu, "v_min_u32", 64)
  RUN_VALU(k_minu_sdwa, "v_min_u32_sdwa src1 WORD_0", 64)
  RUN_VALU(k_mul24, "v_mul_u32_u24", 64)
  RUN_VALU(k_mul24_sdwa, "v_mul_u32_u24_sdwa src0 WORD_1", 64)
  RUN_VALU(k_addu_sdwa, "v_add_u32_sdwa src0 WORD_1", 64)
  RUN_VALU(k_lshl, "v_lshlrev_b32", 64)
  RUN_VALU(k_lshr, "v_lshrrev_b32", 64)
  RUN_VALU(k_and, "v_and_b32", 64)
  RUN_VALU(k_xor, "v_xor_b32", 64)
  RUN_VALU(k_subu, "v_sub_u32", 64)
  RUN_VALU(k_maxf, "v_max_f32", 64)
  RUN_VALU(k_subf, "v_sub_f32", 64)
  RUN_VALU(k_fmac, "v_fmac_f32", 64)
  RUN_VALU(k_mov, "v_mov_b32", 64)
  RUN_VALU(k_cmp, "v_cmp_le_u32 vcc", 64)
  RUN_VALU(k_bfe, "v_bfe_u32", 64)
  RUN_VALU(k_andor, "v_and_or_b32", 64)
  RUN_VALU(k_madu16, "v_mad_u32_u16 op_sel hi", 64)
  RUN_VALU(k_dot2, "v_dot2_u32_u16", 64)
  RUN_VALU(k_pkaddu16, "v_pk_add_u16", 64)
  RUN_VALU(k_pkminu16, "v_pk_min_u16", 64)
  RUN_VALU(k_mullo, "v_mul_lo_u32", 64)
  RUN_VALU(k_mulden, "v_mul_f32 by a denormal constant", 64)
  RUN_VALU(k_cvtu, "v_cvt_u32_f32", 64)
  RUN_VALU(k_rndne, "v_rndne_f32", 64)
  RUN_VALU(k_max3u, "v_max3_u32", 64)
  RUN_VALU(k_addlshl, "v_add_lshl_u32", 64)
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = cus * wps;
    report("v_add_f32 sgpr operand", wps, double(kIters) * 64,
           run([&] { hipLaunchKernelGGL(k_add_sgpr, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 1.5f); }, blocks, d_cyc));
  }
  {
    float4* d_table;
    CHECK(hipMalloc(&d_table, 4096));
    CHECK(hipMemset(d_table, 0, 4096));
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      report("s_load_dwordx4 [cyc per wave-instr per SIMD]", wps, double(kIters) * 8,
             run([&] { hipLaunchKernelGGL(k_sload, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, d_table); }, blocks, d_cyc));
    }
  }
  for (float spread : {0.1f, 0.3f, 0.6f, 0.9f}) {
    for (int layout = 0; layout < 3; ++layout) {
      const int sx = layout == 0 ? 24 : layout == 1 ? 26 : 34;
      const int sxy = layout == 0 ? 24 * 24 : layout == 1 ? 26 * 25 : 34 * 21;
      for (int wps : {2, 4}) {
        const int blocks = cus * wps;
        char nm[160];
        snprintf(nm, sizeof nm, "ds_read_u16 rotation-lane box gather spread %.1f strides (%d,%d) [cyc per wave-instr per CU]", spread, sx, sxy);
        Result r = run([&] { hipLaunchKernelGGL(k_lds_box, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, spread, sx, sxy); }, blocks, d_cyc);
        r.max_cyc /= 4.0;
        r.wall_ms /= 4.0;
        report(nm, wps, double(kIters) * 8, r);
      }
    }
  }
  if (quick) {
    printf("\n]}\n");
    return 0;
  }
  RUN_VALU(k_fma, "v_fma_f32", 64)
  RUN_VALU(k_add, "v_add_f32", 64)
  RUN_VALU(k_mul, "v_mul_f32", 64)
  RUN_VALU(k_pkfma, "v_pk_fma_f32", 64)
  RUN_VALU(k_add3, "v_add3_u32", 64)
  RUN_VALU(k_addu, "v_add_u32", 64)
  RUN_VALU(k_cvtflr, "v_cvt_flr_i32_f32", 64)
  RUN_VALU(k_cvti, "v_cvt_i32_f32", 64)
  RUN_VALU(k_fract, "v_fract_f32", 64)
  RUN_VALU(k_floor, "v_floor_f32", 64)
  RUN_VALU(k_min3, "v_min3_f32", 64)
  RUN_VALU(k_med3, "v_med3_f32", 64)
  RUN_VALU(k_lshladd, "v_lshl_add_u32", 64)
  RUN_VALU(k_mad24, "v_mad_u32_u24", 64)
  RUN_VALU(k_cmpcnd, "v_cmp_ge_f32+v_cndmask (pair)", 64)
  for (int mode : {0, 1}) {
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      std::string nm = std::string("ds_read_b32 ") + (mode == 0 ? "broadcast" : "lane*4");
      // LDS is a per-CU resource: report cycles per wave-instruction per CU (4 SIMDs issue into it)
      Result r = run([&] { hipLaunchKernelGGL(k_lds<32>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, mode); }, blocks, d_cyc);
      r.max_cyc /= 4.0;
      r.wall_ms /= 4.0;
      report((nm + " [cyc per wave-instr per CU]").c_str(), wps, double(kIters) * 8, r);
    }
  }
  for (int mode : {0, 3, 2, 4}) {
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      const char* m = mode == 0 ? "broadcast" : mode == 3 ? "3^3 spread" : mode == 2 ? "5^3 spread" : "9^3 spread";
      std::string nm = std::string("ds_read_u16 ") + m + " [cyc per wave-instr per CU]";
      Result r = run([&] { hipLaunchKernelGGL(k_lds<16>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, mode); }, blocks, d_cyc);
      r.max_cyc /= 4.0;
      r.wall_ms /= 4.0;
      report(nm.c_str(), wps, double(kIters) * 8, r);
    }
  }
  const char* modes[] = {"one address", "1 line", "4 lines", "16 lines", "64 lines", "coalesced"};
  for (int width : {16, 32}) {
    for (int mode = 0; mode < 6; ++mode) {
      for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;
        std::string nm = std::string(width == 16 ? "global_load_ushort " : "global_load_dword ") + modes[mode] +
                         " [cyc per wave-instr per CU]";
        Result r;
        if (width == 16)
          r = run([&] { hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, d_buf, mode); }, blocks, d_cyc);
        else
          r = run([&] { hipLaunchKernelGGL(k_gather<32>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, d_buf, mode); }, blocks, d_cyc);
        r.max_cyc /= 4.0;
        r.wall_ms /= 4.0;
        report(nm.c_str(), wps, double(kIters / 4) * 8, r);
      }
    }
  }
  printf("\n]}\n");
  return 0;
}
