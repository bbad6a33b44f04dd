// Internal: helpers shared by the one-pass kernels on the LDS-DMA ring (fused2.hip, fused3.hip):
// inline-asm LDS-DMA, counted waits, asm MFMAs with tied accumulators and their wait states.
#pragma once
#include "device.hpp"

namespace cofactor {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

namespace onepass {

constexpr int F2_THREADS = 256;
constexpr int TR = FUSED_TILE_ROWS;        // 256 rows per tile
constexpr int COLB = 1040;                 // bytes of one column in a ring slot: 1 KiB + 16 (bank spread, 16-B aligned)
constexpr int PST = 144;                   // bytes of one piece column in a wave's scratch: 64 rows bf16 + 16
constexpr int CST = 64;                    // bytes of one code column in a wave's scratch: 64 rows u8
constexpr int S_FLUSH_TILES = 32;          // a per-key fp32 cell holds <= 32 x 64 adds of bf16 pieces between folds
constexpr int G_FLUSH_TILES = 4;           // Gram chains: <= 64 fp32 adds between fp64 folds (x rows per MFMA)
constexpr unsigned NO_CODE = 16u;          // key not in the dictionary / no such column
constexpr unsigned ROW_OFF = 17u;          // every column of a row the row filter dropped
constexpr int DIRECT_KEYS = 256;
constexpr int DIRECT_STRIDE = 260;

// modes (template parameter MODE)
constexpr int F2_PAIRS = 1, F2_SSUM = 2;   // pair tables; per-key sums of the numeric columns (key counts always)
constexpr int F2_SUB = 4;                  // sub-launch (device.hpp: F2Sub): no Gram, tables land at sub's offsets

struct F2Carve {          // byte offsets into the dynamic LDS block
  int ring, slot_bytes, zero, scratch, scratch_bytes, s, cnt, direct, slot, dcode, total;
};

__device__ __forceinline__ unsigned fhash2(int32_t key, int cap) {
  return ((unsigned)key * 0x9E3779B1u) >> (32 - (31 - __builtin_clz(cap)));
}

// code of one key in the LDS copy of a dictionary (NO_CODE if absent)
__device__ __forceinline__ unsigned lds_lookup1(const unsigned long long *slots, const int32_t *codes, int cap,
                                               unsigned key) {
  const unsigned long long want = (1ull << 32) | (unsigned long long)key;
  unsigned h = fhash2((int32_t)key, cap);
  for (int probe = 0; probe < cap; probe++) {
    const unsigned long long cur = slots[h];
    if (cur == want) return (unsigned)codes[h] & 0xFFu;
    if (cur == 0ull) break;
    h = (h + 1) & (cap - 1);
  }
  return NO_CODE;
}

// LDS-DMA: 64 lanes x 16 (or 4) bytes from each lane's global address to LDS at lds_dst + 16 (4) x lane.
// As asm statements: hipcc counts a builtin glds as an LDS store that any later LDS access may
// alias and waits vmcnt(0) before the next ds_read, i.e. drains the whole ring every tile.  The
// waits are placed by hand (wait_vmcnt + s_barrier).  M0 is written in the statement that reads it.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: one VGPR (the
// lane offset) serves every column, where per-lane 64-bit addresses cost two VGPRs per column (and,
// once spilled, a scratch reload + s_waitcnt vmcnt(0) in front of every DMA: the whole ring drained).
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4_s(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// s_waitcnt vmcnt(k) for a wave-uniform k (the immediate has to be a constant)
__device__ __forceinline__ void wait_vmcnt(int k) {
  switch (k) {
#define W(K) case K: wait_vmcnt_imm<K>(); break;
    W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19)
    W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37)
    W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47) W(48)
#undef W
    default: wait_vmcnt_imm<0>(); break;
  }
}

// S-MFMA with its accumulator in arch VGPRs.  hipcc puts every builtin MFMA's accumulator into the
// 256 AGPRs when a kernel may use the whole register file (and spills beyond them); the pair
// blocks fill those, so the per-key-sum blocks live on the VGPR side through this statement.
// Hazards hipcc does not see for an asm MFMA: (i) an operand register written by the VALU
// instruction right before it is read stale (measured: the second of two back-to-back
// v_perm + MFMA groups computed with the first group's operand) -> wait states in the statement;
// (ii) its result is only read in flush_s, behind mfma_settle().
__device__ __forceinline__ void smfma(f32x4 &acc, u32x4 a, u32x4 b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// int8 one-hot bytes (0x40) of 8 rows -> bf16 one-hot operand (0x4000 = 2.0): one byte shuffle per
// two rows.  The wait states at the end let the MFMA that follows read the last v_perm's result.
__device__ __forceinline__ u32x4 onehot_bf16(unsigned w0, unsigned w1) {
  u32x4 r;
  asm("v_perm_b32 %0, 0, %4, %6\n\tv_perm_b32 %1, 0, %4, %7\n\tv_perm_b32 %2, 0, %5, %6\n\tv_perm_b32 %3, 0, %5, %7\n\ts_nop 2"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
      : "v"(w0), "v"(w1), "s"(0x010C000Cu), "s"(0x030C020Cu));
  return r;
}
// pair-count MFMA, accumulator tied in the AGPRs (left to itself hipcc gives most of these MFMAs a
// destination different from srcC and copies 180 registers back every tile)
__device__ __forceinline__ unsigned xad(unsigned a, unsigned b, unsigned c) {   // (a ^ b) + c in one VALU op
  unsigned r;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}
template <int M>
__device__ __forceinline__ void settle_operands(i32x4 (&oh)[M]) {
  static_assert(M == 2 || M == 4 || M == 6 || M == 8 || M == 10, "M is even, <= 10");
  if constexpr (M == 2) asm volatile("s_nop 3" : "+v"(oh[0]), "+v"(oh[1]));
  if constexpr (M == 4) asm volatile("s_nop 3" : "+v"(oh[0]), "+v"(oh[1]), "+v"(oh[2]), "+v"(oh[3]));
  if constexpr (M == 6) asm volatile("s_nop 3" : "+v"(oh[0]), "+v"(oh[1]), "+v"(oh[2]), "+v"(oh[3]), "+v"(oh[4]), "+v"(oh[5]));
  if constexpr (M == 8)
    asm volatile("s_nop 3" : "+v"(oh[0]), "+v"(oh[1]), "+v"(oh[2]), "+v"(oh[3]), "+v"(oh[4]), "+v"(oh[5]), "+v"(oh[6]), "+v"(oh[7]));
  if constexpr (M == 10)
    asm volatile("s_nop 3" : "+v"(oh[0]), "+v"(oh[1]), "+v"(oh[2]), "+v"(oh[3]), "+v"(oh[4]), "+v"(oh[5]), "+v"(oh[6]), "+v"(oh[7]),
                 "+v"(oh[8]), "+v"(oh[9]));
}
__device__ __forceinline__ void pmfma(i32x4 &acc, i32x4 a, i32x4 b) {
  asm("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// Wait states between an asm MFMA and the first read of its result (hipcc's hazard recognizer does
// not see through the asm statement).  The operand ties the wait to THAT accumulator's last MFMA.
__device__ __forceinline__ void mfma_settle(f32x4 &acc) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc)); }
__device__ __forceinline__ void mfma_settle(i32x4 &acc) { asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc)); }
// Pair-count MFMA with its accumulator tied in ARCH VGPRs (fused3: at two waves per SIMD hipcc
// splits the 256 registers of a wave 128 / 128 as soon as any AGPR is named, and 180 pair
// registers do not fit 128; a kernel that names no AGPR gets all 256 as arch VGPRs).
__device__ __forceinline__ void pmfma_v(i32x4 &acc, i32x4 a, i32x4 b) {
  asm("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_settle_v(i32x4 &acc) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc)); }

}  // namespace onepass
}  // namespace cofactor
