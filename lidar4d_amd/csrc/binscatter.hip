// Sorted gradient scatter for the large 3-D hash tables (static grid 2^19 x 4, flow grid 2^18 x 8) on gfx950.
//
// Scattered global fp32 atomics sustain ~20 G lane-ops/s on MI355X (profiles/r01_ubench_global_atomics.txt);
// a 16,384-ray step needs 3.2 G of them for the static grid alone (148 ms measured).  Tables of 8 MB per level do
// not fit LDS, so contributions are first *binned* by table segment and then reduced segment by segment in LDS:
//
//   pass 1  (one thread per (sample, level)): the 2^D corner contributions {entry, w * g[0..NV)} (fp16 payload) are
//           partitioned inside the workgroup by bin = entry >> shift (LDS histogram + ranks) and written, sorted by
//           bin, into the workgroup's own fixed slot of the record buffer together with its bin offsets -- no global
//           atomics, no overflow, deterministic layout.  Coarse levels first merge runs of samples in one cell along the
//           ray (DPP row scan, wave_dev.h).  Blocks are ordered level-fast and XCD-aware so that the levels of a tile
//           share the gradient rows in one L2.
//   pass 2  (one workgroup per (level, bin)): walks every pass-1 workgroup's run for its bin (16 lanes per run),
//           accumulates in LDS as int64 fixed point (ds_add_u64: 2.9 T lane-ops/s, exact and order independent), and
//           adds the segment to the fp32 gradient table with plain stores -- each segment has exactly one owner.
//
// Non-hashed (dense, coarse) levels fall back to run-reduced global atomics.
//
// PAIR RECORDS.  The coherent-prime hash leaves the first coordinate unmultiplied (index = x ^ y p1 ^ z p2, masked), so the
// two x-neighbours of a cell sit in entries i and i ^ m with m = x ^ (x + 1) = 2^(tz + 1) - 1 (tz = trailing ones of x): in
// the same bin whenever tz + 1 <= shift, i.e. all but 2^-shift of the time.  On the fine levels -- no two samples of a wave
// share a cell, nothing to merge -- one record therefore carries BOTH: key word = [bin-local index : 13 | tz : 4 | fx : 15],
// payload = w_yz * g (fp16); pass 2 adds payload * (1 - fx) to entry i and payload * fx to entry i ^ m.  Half the records, half
// the bytes written and read back, half the ranking work; the weights lose nothing (fx to 2^-15, the payload is rounded to
// fp16 once, as before).  tz code 15 = a single record: merged runs of the coarse levels, and the two halves of a pair that
// straddles two bins (tz + 1 > shift: one cell in 2^shift -- but a ray that runs nearly perpendicular to x stays in such a
// cell for dozens of samples, so bursts of them are an everyday event).  A workgroup's slot therefore keeps room for 2^D
// records per lane -- no case in which records do not fit, nothing that depends on atomic arrival -- while only the records
// that exist (2^(D-1) per lane, typically) are written and read.
#define BS_CODE_SINGLE 15u
#define BS_KEY_BITS 13
#define BS_FX_ONE 32767.0f
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "hashgrid_dev.h"
#include "wave_dev.h"
#include "binscatter.h"

#define BS_THREADS 512
#define BS_MAX_BINS 256
#define BS_GROUP 8
#define BS_XCD_BINS 1
#define BS_UNROLL 4
#define BS_MIN_WAVES 4  // pass 1: wavefronts per SIMD the register allocation must leave room for (128 registers)

template <int NV>
struct RecWords { static constexpr int n = 1 + (NV + 1) / 2; };  // key + packed halfs

template <int NV>
__device__ __forceinline__ void pack_payload(const float v[NV], uint32_t out[(NV + 1) / 2]) {
#pragma unroll
  for (int q = 0; q < (NV + 1) / 2; ++q) {
    const half2_t h = {f2h_grad(v[2 * q]), 2 * q + 1 < NV ? f2h_grad(v[2 * q + 1]) : (half_t)0.0f};
    out[q] = __builtin_bit_cast(uint32_t, h);
  }
}

template <int D, int NV>
__global__ void __launch_bounds__(BS_THREADS, BS_MIN_WAVES) bin_pass1_kernel(GridDesc desc, const float* __restrict__ x, int64_t P, int x_stride,
                                                              BsCols cols, const half_t* __restrict__ g, int g_stride, int g_col,
                                                              float pre_scale, int shift, int64_t n_wg,
                                                              uint16_t* __restrict__ offs, uint32_t* __restrict__ bins,
                                                              float* __restrict__ lvl_max, float* __restrict__ out, float out_scale) {
  constexpr int NC = 1 << D;
  constexpr int NW = RecWords<NV>::n;
  // ONE histogram per workgroup: a rank is the value the returning LDS atomic hands back.  (Earlier: one histogram per wavefront
  // plus a prefix pass over the wavefronts -- needed when 8 records per lane of a 1024-thread workgroup went into <= 128 bins and
  // piled dozens of same-address atomics onto every counter.  With <= 4 records per lane into 64-256 hashed, i.e. random, bins a
  // wave-level atomic meets 2-4 equal addresses, and the prefix pass, its barrier and 7/8 of the zeroing were pure overhead per
  // level.)  The order of the records inside a run now depends on atomic arrival; pass 2 sums in integers, so nothing downstream
  // depends on it.
  __shared__ uint32_t hist[BS_MAX_BINS], boff[BS_MAX_BINS + 1];
  // typically NC / 2 records per lane: NC / 2 pair records, or NC single records per run with <= 32 runs per wave
  constexpr uint32_t CAP = BS_THREADS * NC;  // records per (workgroup, level) slot: every pair of every lane may have to be split
  __shared__ __attribute__((aligned(16))) uint32_t stage[CAP * NW];
  __shared__ uint32_t total_s;
  // per-level maximum |g| of this workgroup (bit patterns of non-negative floats: ordered like unsigned integers).  Updating the
  // global lvl_max[] per wavefront and level meant a load of the current maximum and a wait for it -- one exposed L2 round trip
  // per level in every wavefront of the workgroup at the same moment, with nothing else to run (two workgroups per CU).
  __shared__ uint32_t lmax_s[L4D_MAX_LEVELS];
  // One workgroup walks ALL levels of its tile of samples.  (Earlier: one workgroup per (tile, level), ordered level-fast
  // and XCD-aware so that the levels of a tile at least met in one L2.  Every such workgroup started with a cold,
  // 256-byte-strided read of its level's 2-8 bytes of the gradient rows and sat out that latency at two workgroups per
  // CU.)  The coordinates are read once, a level's gradient values are fetched while the previous level is being ranked.
  const int n_lv = desc.n_levels;
  const int64_t tile = xcd_tile(blockIdx.x, gridDim.x);
  if (tile >= n_wg) return;  // block-uniform, before any barrier
  const int lane = __lane_id();
  const int64_t pr = tile * blockDim.x + threadIdx.x;
  const bool valid = pr < P;
  const int64_t p = valid ? pr : P - 1;
  float xin[D];
#pragma unroll
  for (int d = 0; d < D; ++d) xin[d] = x[p * x_stride + cols.c[d]];
  const half_t* grow = g + p * g_stride + g_col;
  // All levels' gradient values of the sample are fetched ONCE, as whole 16-byte pieces, when they fit 64 bytes (L x NV <= 32
  // halfs): read level by level -- 2 to 8 bytes out of a 256-byte-strided row per level phase -- every level pulled its own
  // 64-byte sector across the fabric (PMC: 5.4 GB fetched for 0.8 GB of values).  The level loop is not unrolled, so the
  // level's dwords are picked by a uniform index into a register VECTOR (an indexed array would live in scratch).
  // Held 8 dwords (32 bytes of the row) at a time: the next 32 bytes are fetched while the last level of the current ones is
  // ranked.  (All 64 bytes up front cost 8 more registers for the whole kernel, which now sits at the 128-register limit of its
  // four wavefronts per SIMD.)
  constexpr int GW = 8;
  const bool g_in_regs = n_lv * NV <= 32 && (g_stride * 2) % 16 == 0 && (g_col * 2) % 16 == 0;  // block-uniform
  typedef uint32_t GwVec __attribute__((ext_vector_type(GW)));  // a vector, so that a uniform index becomes relative VGPR addressing
  GwVec gw;
  // (g_in_regs only.  The loads are unconditional -- a piece behind the last level re-reads piece 0 and is never picked: a load
  // under a condition merges with its default value in register copies, and the copies wait for the load where it is issued)
  auto gw_load = [&](int chunk) {  // dwords 8 chunk .. 8 chunk + 7 of the row's gradient columns
#pragma unroll
    for (int q = 0; q < GW / 4; ++q) {
      const int piece = (chunk * 2 + q) * 8 < n_lv * NV ? chunk * 2 + q : 0;  // block-uniform
      const uint4 u = *reinterpret_cast<const uint4*>(grow + piece * 8);
      gw[4 * q + 0] = u.x; gw[4 * q + 1] = u.y; gw[4 * q + 2] = u.z; gw[4 * q + 3] = u.w;
    }
  };
  gw = GwVec{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  if (g_in_regs) gw_load(0);
  auto gw_pick = [&](int k) -> uint32_t { return gw[k & (GW - 1)]; };
  half_t gnext[NV];
  for (int i = threadIdx.x; i < BS_MAX_BINS; i += BS_THREADS) hist[i] = 0;
  if (threadIdx.x < L4D_MAX_LEVELS) lmax_s[threadIdx.x] = 0u;
  __syncthreads();
  // The level loop exists twice (a generic lambda over GREG = "gradient dwords in the register window"): the loads of the other
  // shapes' path, in ONE loop with the common path, made the compiler wait for every outstanding memory operation at the join --
  // the gradient dwords just requested and the previous level's copy-out stores included -- on the common path as well.
  // COPY-OUT, ONE LEVEL LATE.  A level's staged records (sorted by bin; LDS -> the workgroup's slot, 16 bytes per lane: the slot is
  // 16-byte aligned and large enough for the rounded-up tail) are stored at the HEAD of the next level, behind the pick of that
  // level's gradient dwords.  The pick waits for the dwords requested a level earlier, and the compiler's wait there is for every
  // outstanding memory operation: with the stores issued at a level's end it drained them at the head of the next one, eight
  // times per workgroup, with nothing else to run at two workgroups per CU.  Issued behind the pick they have a whole level
  // (three barriers) to complete before the next wait.  Nothing touches `stage` or total_s before the next level's first
  // barrier, which every wavefront reaches only behind its share of the copy.
  int copy_lvl = -1;  // block-uniform
  auto copy_out = [&]() {
    if (copy_lvl < 0) return;
    const uint32_t total = total_s;
    uint32_t* dst = bins + ((uint64_t)copy_lvl * n_wg + tile) * (uint64_t)(CAP * NW);
    const uint32_t n16 = (total * NW + 3) >> 2;
    for (uint32_t q = threadIdx.x; q < n16; q += blockDim.x) reinterpret_cast<uint4*>(dst)[q] = reinterpret_cast<const uint4*>(stage)[q];
    copy_lvl = -1;
  };
  auto levels = [&](auto greg_tag) {
  constexpr bool GREG = decltype(greg_tag)::value;
  for (int lvl = 0; lvl < n_lv; ++lvl) {
  const bool hashed = (desc.hashed_mask >> lvl) & 1u;
  const uint32_t size = desc.size[lvl];
  const int nbins = (int)((size + (1u << shift) - 1) >> shift);
  // (binned levels: hashed with a power-of-two table -- what every hashed level of a tiny-cuda-nn grid is; anything else takes the
  // atomic path below -- so that their index is the plain xor-and-mask of hashgrid_dev.h grid_index_fast)
  const bool binned = hashed && is_pow2(size) && nbins <= BS_MAX_BINS && nbins > 1;

  if (!GREG) {  // (shapes outside the register window: this level's values, loaded where they are used)
#pragma unroll
    for (int j = 0; j < NV; ++j) gnext[j] = grow[lvl * NV + j];
  }
  if (GREG) {  // this level's NV halfs out of the preloaded dwords
    uint32_t w[(NV + 1) / 2];
#pragma unroll
    for (int q = 0; q < (NV + 1) / 2; ++q) w[q] = gw_pick(NV == 1 ? lvl >> 1 : lvl * (NV / 2) + q);
    // (picked before the next dwords are requested INTO the same registers: loads scheduled in front of the pick need a second
    // register set and a copy behind them -- which waits for them on the spot)
#pragma unroll
    for (int q = 0; q < (NV + 1) / 2; ++q) asm volatile("" : "+v"(w[q]) : : "memory");
    if (NV == 1 && (lvl & 1)) w[0] >>= 16;
    const half_t* hw = reinterpret_cast<const half_t*>(w);
#pragma unroll
    for (int j = 0; j < NV; ++j) gnext[j] = hw[j];
    // first dword of the next level (NV = 1: two levels per dword); at a 32-byte boundary the held dwords are all consumed
    const int k_next = NV == 1 ? (lvl + 1) >> 1 : (lvl + 1) * (NV / 2);
    if (lvl + 1 < n_lv && k_next % GW == 0 && (NV > 1 || ((lvl + 1) & 1) == 0)) gw_load(k_next / GW);
  }
  copy_out();  // the previous level's records (see above)
  float gv[NV];
  bool any = false;
  float amax = 0.0f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    gv[j] = valid ? h2f(gnext[j]) * pre_scale : 0.0f;
    any |= gv[j] != 0.0f;
    amax = amax_nf(amax, gv[j]);
  }
  // (No barrier here: the histogram was left zeroed by the scan of the previous binned level, and this level's staging writes and
  // bin offsets come two barriers down, behind every wavefront's copy-out of the previous level -- three barriers per level, not five.)

  Cell<D> c = locate<D>(xin, desc.scale[lvl]);
  uint32_t keys[NC];
  float vals[NC][NV];
  bool emit[NC];
  uint32_t pos[NC];
  const bool wave_any = __any(any);
  // Consecutive lanes are consecutive samples of a ray: on coarse levels several of them sit in one cell and hit the same
  // 2^D entries.  Binned levels merge such runs inside 16-lane rows with the DPP scan (one record per run and corner);
  // the dense fallback keeps the wave-wide shuffle reduction (global atomics are the expensive resource there).
  int n_heads = 64;
  RowRuns runs;
  bool use_scan = false;
  // (fine levels: the 64 consecutive samples of a wavefront cross far more than 32 cells, no two of them share one -- the run
  // detection below (three DPP compares, a ballot, the run bookkeeping: ~45 instructions) is skipped when the first and the last
  // lane are more than 48 cells apart along some axis; a heuristic about WORK only, the pair path is always correct)
  bool try_merge = binned && wave_any;
  if (try_merge) {
    bool same = true;  // same cell as the previous lane (the first lane of a row never is: old = ~cell, bound_ctrl off)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp((int)~c.cell[d], (int)c.cell[d], 0x111, 0xf, 0xf, false);
      same &= pv == c.cell[d];
    }
    // key = running id that changes exactly where the cell changes (lanes without a gradient carry zeros: harmless)
    const unsigned long long brk = __ballot(!same);
    runs = row_runs((uint32_t)__popcll(brk & (~0ull >> (63 - lane))), &n_heads);
    use_scan = n_heads <= 32;  // wave-uniform: merged runs emit single records, at most NC per run = NC / 2 per lane
  }
  const bool pairs = binned && !use_scan;  // wave-uniform
  uint32_t fxq = 0u;  // keys[] carry the record's tz code in bits 24..27 from here on (entries per level <= 2^24: checked by the host side)
  if (pairs) {
    // one record per x-neighbour pair (corners 2q, 2q + 1): slots 0 .. NC/2 - 1 are used, the rest stay silent
    fxq = (uint32_t)fx_round(c.frac[0] * BS_FX_ONE);
    const float fx = (float)fxq * (1.0f / BS_FX_ONE);
#pragma unroll
    for (int q = 0; q < NC / 2; ++q) {
      uint32_t g0[D], g1[D];
      const float w0 = corner<D>(c, 2 * q, g0);
      (void)corner<D>(c, 2 * q + 1, g1);
      float wyz = 1.0f;  // weight without the x factor
#pragma unroll
      for (int d = 1; d < D; ++d) wyz *= ((2 * q) >> d) & 1 ? c.frac[d] : 1.0f - c.frac[d];
      (void)w0;
      const uint32_t k0 = grid_index_fast<D>(g0, size - 1u), k1 = grid_index_fast<D>(g1, size - 1u);
      const uint32_t m = k0 ^ k1;
      const bool paired = (m & (m + 1u)) == 0u && m != 0u && (m >> shift) == 0u && __popc(m) <= (int)BS_CODE_SINGLE;
      keys[q] = k0 | ((uint32_t)(__popc(m) - 1) << 24);
      float* v = vals[q];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = wyz * gv[j];
      bool nz = false;
#pragma unroll
      for (int j = 0; j < NV; ++j) nz |= v[j] != 0.0f;
      emit[q] = any && nz;
      // straddles two bins (2^-shift of the pairs): two single records, the neighbour's in the otherwise unused slot q + NC/2
      const bool split = emit[q] && !paired;
      emit[q + NC / 2] = split;
      keys[q + NC / 2] = k1 | (BS_CODE_SINGLE << 24);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        vals[q + NC / 2][j] = v[j] * fx;
        if (split) v[j] *= 1.0f - fx;
      }
      if (split) keys[q] = k0 | (BS_CODE_SINGLE << 24);
    }
  } else {
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    uint32_t gg[D];
    const float w = corner<D>(c, k, gg);
    keys[k] = binned ? grid_index_fast<D>(gg, size - 1u) : grid_index<D>(gg, desc.res[lvl], size, hashed);
#pragma unroll
    for (int j = 0; j < NV; ++j) vals[k][j] = w * gv[j];
    if (binned) {
      emit[k] = any;
      if (use_scan) {
        row_scan<NV>(runs, vals[k]);
        emit[k] = runs.tail;
      }
    } else {
      emit[k] = wave_any ? wave_run_reduce<NV>(keys[k], any, vals[k]) : false;
    }
    if (emit[k]) {
      bool nz = false;
#pragma unroll
      for (int j = 0; j < NV; ++j) nz |= vals[k][j] != 0.0f;
      emit[k] = nz;
    }
    if (binned) keys[k] |= BS_CODE_SINGLE << 24;
  }
  }
  if (!binned) {  // dense / tiny level: run-reduced atomics straight into the output (block-uniform branch)
    float* o = out + (size_t)desc.offset[lvl] * NV;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (emit[k]) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
          if (vals[k][j] != 0.0f) atomicAdd(o + (size_t)keys[k] * NV + j, vals[k][j] * out_scale);
      }
    continue;  // block-uniform: next level
  }
  amax = wave_max(amax);
  if (lane == 0 && amax > 0.0f) atomicMax(&lmax_s[lvl], __float_as_uint(amax));  // (ds_max_u32, nothing returned: no wait)

  // In pair mode the upper half of the slots only holds the second halves of pairs that straddle two bins (one pair in 2^shift):
  // a wavefront without one skips their ranking and staging code altogether (wave-uniform branch instead of exec-masked no-ops).
  bool upper = true;
  // rank inside the workgroup
#pragma unroll
  for (int k = 0; k < NC / 2; ++k) pos[k] = emit[k] ? atomicAdd(&hist[(keys[k] & 0xFFFFFFu) >> shift], 1u) : 0u;
  if (upper) {
#pragma unroll
    for (int k = NC / 2; k < NC; ++k) pos[k] = emit[k] ? atomicAdd(&hist[(keys[k] & 0xFFFFFFu) >> shift], 1u) : 0u;
  }
  __syncthreads();
  const bool scan_io = true;
  if (threadIdx.x < 64) {  // exclusive scan of the bin totals by one wave: BPL consecutive bins per lane
    constexpr int BPL = BS_MAX_BINS / 64;
    uint32_t c[BPL], sum = 0;
#pragma unroll
    for (int q = 0; q < BPL; ++q) {
      const int b = lane * BPL + q;
      c[q] = b < nbins ? hist[b] : 0u;
      if (b < nbins) hist[b] = 0u;  // ready for the next level (every rank of this level has been handed out: barrier above)
      sum += c[q];
    }
    // inclusive prefix over the wavefront with DPP (row scan, then the row totals carried upwards: six VALU instructions; as in
    // pass 2).  Six ds_bpermute round trips here were ~800 cycles of the ONE wavefront that the workgroup's other fifteen wait for.
    uint32_t inc = sum;
#define L4D_ADD_DPP(ctrl, rmask) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, ctrl, rmask, 0xf, true)
    L4D_ADD_DPP(0x111, 0xf);  // row_shr:1 (bound_ctrl: lanes without a source add 0)
    L4D_ADD_DPP(0x112, 0xf);
    L4D_ADD_DPP(0x114, 0xf);
    L4D_ADD_DPP(0x118, 0xf);
    L4D_ADD_DPP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    L4D_ADD_DPP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef L4D_ADD_DPP
    // Bin offsets are stored TRANSPOSED, offs[level][bin][workgroup]: pass 2 walks one bin over all workgroups, and read from
    // a [workgroup][bin] table every run cost it a 64-byte sector for two 2-byte numbers -- as many fabric requests as the run's
    // records themselves.  The 2-byte stores below land in lines that the neighbouring tiles (same XCD, dispatched together:
    // xcd_tile) complete within the same L2.
    const uint32_t nwg32 = (uint32_t)n_wg;  // (levels x bins x workgroups < 2^32: checked by the host side)
    // scalar base + 32-bit byte offset per lane: as 64-bit per-lane addresses the level-invariant part was hoisted out of the level
    // loop into two register pairs that did not fit and were reloaded from scratch here -- in the one wavefront that the other
    // fifteen of the workgroup wait for
    const uint64_t ob = reinterpret_cast<uint64_t>(offs + (uint64_t)lvl * (BS_MAX_BINS + 1) * n_wg + tile);
    typedef __attribute__((address_space(1))) char GlobalByte;  // (an integer cast to a plain pointer is a FLAT address)
    typedef __attribute__((address_space(1))) uint16_t GlobalU16;
    GlobalByte* o = (GlobalByte*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ob >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ob));
    uint32_t excl = inc - sum;
    uint32_t lane_off = (uint32_t)(lane * BPL) * nwg32 * 2u;
    asm volatile("" : "+v"(lane_off));  // (opaque per level: hoisted out of the level loop the offsets become 64-bit register pairs again)
#pragma unroll
    for (int q = 0; q < BPL; ++q) {
      const int b = lane * BPL + q;
      boff[b] = excl;
      if (scan_io && b <= nbins) *(GlobalU16*)(o + (lane_off + (uint32_t)q * nwg32 * 2u)) = (uint16_t)excl;
      excl += c[q];
    }
    if (scan_io && lane == 63) {
      total_s = inc;
      boff[BS_MAX_BINS] = inc;
      if (nbins == BS_MAX_BINS) *(GlobalU16*)(o + (uint32_t)BS_MAX_BINS * nwg32 * 2u) = (uint16_t)inc;
    }
  }
  __syncthreads();
  auto stage_slot = [&](int k) {
    if (emit[k]) {
      const uint32_t b = (keys[k] & 0xFFFFFFu) >> shift;
      const uint32_t r = boff[b] + pos[k];
      const uint32_t code = keys[k] >> 24;
      stage[r * NW] = (keys[k] & ((1u << shift) - 1u)) | (code << BS_KEY_BITS) | (code == BS_CODE_SINGLE ? 0u : fxq << (BS_KEY_BITS + 4));
      uint32_t pay[NW - 1];
      pack_payload<NV>(vals[k], pay);
#pragma unroll
      for (int q = 0; q < NW - 1; ++q) stage[r * NW + 1 + q] = pay[q];
    }
  };
#pragma unroll
  for (int k = 0; k < NC / 2; ++k) stage_slot(k);
  if (upper) {
#pragma unroll
    for (int k = NC / 2; k < NC; ++k) stage_slot(k);
  }
  __syncthreads();
  copy_lvl = lvl;  // the staged records leave at the head of the next level (or behind the loop)
  }  // levels
  };
  if (g_in_regs) levels(std::true_type{});
  else levels(std::false_type{});
  copy_out();
  __syncthreads();
  if ((int)threadIdx.x < n_lv) {
    const uint32_t m = lmax_s[threadIdx.x];
    if (m != 0u) atomic_max_nonneg(lvl_max + threadIdx.x, __uint_as_float(m));
  }
}

// CSHIFT > 0: the bin size is a compile-time constant, so the [value][entry] accumulator addresses are ONE register (entry * 8) plus
// immediate offsets (j * 2^CSHIFT * 8 <= 48 KB fits the DS offset field) -- the kernel issues ~100 VALU instructions per record on 8
// wavefronts per SIMD and is bound by exactly those (SQ_ACTIVE_INST_VALU: 11 % of every wave's cycles x 8 waves).  CSHIFT = 0: run-time.
template <int D, int NV, int CSHIFT = 0>
__global__ void __launch_bounds__(1024) bin_pass2_kernel(GridDesc desc, int shift_rt, int n_wg, int64_t P,
                                                       const uint16_t* __restrict__ offs, const uint32_t* __restrict__ bins,
                                                       const float* __restrict__ lvl_max, float* __restrict__ out, float out_scale) {
  constexpr int NC = 1 << D;
  constexpr int NW = RecWords<NV>::n;
  extern __shared__ long long acc[];
  const int shift = CSHIFT > 0 ? CSHIFT : shift_rt;
  // XCD-aware bin order: workgroup x runs on XCD x % 8, so bin = (x % 8) * (n / 8) + x / 8 gives every XCD a contiguous eighth of a
  // level's bins, and the workgroups it holds at any time own ADJACENT bins.  They walk the pass-1 workgroups in the same order at
  // about the same pace, and a pass-1 slot is sorted by bin: what one of them misses in L2, its neighbours hit (the lines at the run
  // boundaries, the other half of every 128-byte request).  With bin = x those neighbours sat in eight different L2s.
  const int lvl = blockIdx.y;
  const int b = (BS_XCD_BINS && gridDim.x % 8 == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const uint32_t size = desc.size[lvl];
  const bool hashed = (desc.hashed_mask >> lvl) & 1u;
  const int nbins = (int)((size + (1u << shift) - 1) >> shift);
  if (!hashed || !is_pow2(size) || nbins > BS_MAX_BINS || nbins <= 1 || b >= nbins) return;
  const float gmax = lvl_max[lvl];
  if (!(gmax > 0.0f)) return;
  const uint32_t lo = (uint32_t)b << shift;
  if (nonfinite(gmax)) {  // overflowed upstream gradient (inf / nan in g): hand it on to the table gradient
    if (threadIdx.x == 0) out[((size_t)desc.offset[lvl] + lo) * NV] = __builtin_nanf("");
    return;
  }
  const int seg = CSHIFT > 0 ? (1 << CSHIFT) : (1 << shift);  // entries per bin = stride of the [value][entry] accumulator layout
  const int n_ent = (int)min(1u << shift, size - lo);
  const int n_el = n_ent * NV;
  for (int i = threadIdx.x; i < seg * NV; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  // Fixed point: every contribution is |v| <= gmax, scaled to 30 bits and converted with ONE v_cvt_i32_f32 (a float -> int64
  // conversion is a dozen instructions on this ISA, eight of them per record: pass 2 was bound by exactly that), then
  // sign-extended into the 64-bit accumulator -- 2^33 contributions per entry before it could overflow, quantisation 2^-26 of
  // the level's largest gradient (the payload itself carries 11 bits).
  // (a merged run of a coarse level sums up to 16 lanes of one DPP row: 16 gmax bounds every record)
  const float fxs = fx_scale(gmax * 16.5f, 30);
  // groups of BS_GROUP lanes, one pass-1 workgroup's run each.  Every run costs a dependent pair of loads (its offsets,
  // then its records): small groups = many independent chains in flight, which is what hides that latency
  constexpr int NGRP = 1024 / BS_GROUP;
  const int grp = threadIdx.x / BS_GROUP, l16 = threadIdx.x % BS_GROUP;
  auto add = [&](uint32_t w0, const uint32_t* wd) {  // w0: record key word (see PAIR RECORDS above)
    const half_t* hv = reinterpret_cast<const half_t*>(wd);
    const uint32_t local = w0 & ((1u << BS_KEY_BITS) - 1u), code = (w0 >> BS_KEY_BITS) & 15u;
    const bool single = code == BS_CODE_SINGLE;
    const float f1 = single ? 0.0f : (float)(w0 >> (BS_KEY_BITS + 4)) * (1.0f / BS_FX_ONE);
    const float s0 = (1.0f - f1) * fxs, s1 = f1 * fxs;
    const uint32_t other = local ^ ((2u << code) - 1u);
    // accumulators are laid out [value j][entry]: for a given j the 64 lanes of an atomic hit random ENTRIES, i.e. all 64 banks.
    // ([entry][j] put every lane of the instruction on the same NV-th of the banks: 8-way conflicts at NV = 4.)
    unsigned long long* a0 = reinterpret_cast<unsigned long long*>(acc) + local;
    unsigned long long* a1 = reinterpret_cast<unsigned long long*>(acc) + other;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float v = h2f(hv[j]);
      if (v != 0.0f) {  // (many payloads ARE zero: w * g below the smallest fp16 -- dropping this test cost 8 % / 35 % at NV = 4 / 2)
        atomicAdd(a0 + j * seg, (unsigned long long)(long long)fx_round(v * s0));
        if (!single) atomicAdd(a1 + j * seg, (unsigned long long)(long long)fx_round(v * s1));
      }
    }
  };
  // The runs of this bin, one per pass-1 workgroup, consecutive groups take consecutive workgroups: their offsets
  // offs[level][bin][w], offs[level][bin + 1][w] are two dense arrays (fetched one iteration ahead), so the only scattered
  // accesses left are the records themselves.
  const uint16_t* o0 = offs + ((uint64_t)lvl * (BS_MAX_BINS + 1) + b) * n_wg;
  const uint16_t* o1 = o0 + n_wg;
  constexpr uint64_t SLOT = (uint64_t)(BS_THREADS * NC) * NW;
  const uint32_t* lvl_bins = bins + (uint64_t)lvl * n_wg * SLOT;
  // BS_UNROLL runs per group and iteration: their first records are all in flight before any is consumed (each group otherwise has
  // ONE load outstanding -- a dependent chain of ~1.5 us round trips that left the kernel at ~2.8 TB/s of line traffic), and the next
  // iteration's offsets are fetched while this one is processed.
  uint32_t s0n[BS_UNROLL], s1n[BS_UNROLL];
#pragma unroll
  for (int u = 0; u < BS_UNROLL; ++u) {
    const int w = grp + u * NGRP;
    s0n[u] = s1n[u] = 0u;
    if (w < n_wg) { s0n[u] = o0[w]; s1n[u] = o1[w]; }
  }
  for (int w0 = grp; w0 < n_wg; w0 += NGRP * BS_UNROLL) {
    uint32_t s0[BS_UNROLL], s1[BS_UNROLL], key0[BS_UNROLL], wd0[BS_UNROLL][NW - 1];
    const uint32_t* rec[BS_UNROLL];
#pragma unroll
    for (int u = 0; u < BS_UNROLL; ++u) {
      s0[u] = s0n[u];
      s1[u] = s1n[u];
      rec[u] = lvl_bins + (uint64_t)min(w0 + u * NGRP, n_wg - 1) * SLOT;
      const uint32_t r = s0[u] + l16;
      key0[u] = BS_CODE_SINGLE << BS_KEY_BITS;
#pragma unroll
      for (int q = 0; q < NW - 1; ++q) wd0[u][q] = 0u;
      if (r < s1[u]) {  // the run's first BS_GROUP records
        key0[u] = rec[u][r * NW];
#pragma unroll
        for (int q = 0; q < NW - 1; ++q) wd0[u][q] = rec[u][r * NW + 1 + q];
      }
    }
#pragma unroll
    for (int u = 0; u < BS_UNROLL; ++u) {  // the next iteration's offsets
      const int wn = w0 + (BS_UNROLL + u) * NGRP;
      s0n[u] = s1n[u] = 0u;
      if (wn < n_wg) { s0n[u] = o0[wn]; s1n[u] = o1[wn]; }
    }
#pragma unroll
    for (int u = 0; u < BS_UNROLL; ++u) {
      add(key0[u], wd0[u]);  // an absent record carries zeros: no atomics issued
      for (uint32_t r = s0[u] + l16 + BS_GROUP; r < s1[u]; r += BS_GROUP) {
        uint32_t wd[NW - 1];
#pragma unroll
        for (int q = 0; q < NW - 1; ++q) wd[q] = rec[u][r * NW + 1 + q];
        add(rec[u][r * NW], wd);
      }
    }
  }
  __syncthreads();
  const double inv = (double)out_scale / (double)fxs;
  float* o = out + ((size_t)desc.offset[lvl] + lo) * NV;
  for (int i = threadIdx.x; i < n_el; i += blockDim.x) {  // i = entry * NV + j in the table's layout
    const long long v = acc[(i % NV) * seg + i / NV];
    if (v != 0) o[i] += (float)((double)v * inv);  // sole owner of this segment
  }
}

// ---- pass 2, flattened -----------------------------------------------------------------------------------------------------
// bin_pass2_kernel gives every run (one pass-1 tile's records for this bin: 8 on average with pair records, Poisson-distributed)
// to a group of BS_GROUP lanes, and a wavefront runs as many rounds as its LONGEST run needs: with 8 +- 3 records per run nearly
// every wavefront runs two rounds of eight lanes per group for eight records -- half the lanes of every instruction idle in a
// kernel that is bound by its ~100 VALU instructions per record (SQ_ACTIVE_INST_VALU x 8 wavefronts per SIMD = 80 %).
// Here a wavefront takes 64 CONSECUTIVE tiles, reads their 64 + 64 bin offsets with two dense loads, forms the exclusive prefix of
// the run lengths, and walks the concatenation of the 64 runs 64 records at a time: lane j of round q0 handles record q0 + j,
// whichever run it belongs to.  The owner of a record position is found without a search: every tile's lane stamps its lane
// number at the position where its run enters the 64-record window (LDS, one word per position, tagged with the round so that
// nothing is ever cleared), and an inclusive maximum scan over the lanes (six DPP steps) hands every position the last stamp at
// or before it -- runs are laid out in lane order, so that is its owner.  One ds_bpermute brings the owner's (slot start - prefix).
template <int D, int NV, int CSHIFT = 0>
__global__ void __launch_bounds__(1024) bin_pass2_flat_kernel(GridDesc desc, int shift_rt, int n_wg, int64_t P,
                                                            const uint16_t* __restrict__ offs, const uint32_t* __restrict__ bins,
                                                            const float* __restrict__ lvl_max, float* __restrict__ out, float out_scale) {
  constexpr int NC = 1 << D;
  constexpr int NW = RecWords<NV>::n;
  extern __shared__ long long acc[];
  __shared__ uint32_t owner_tag[16][64];  // (not volatile: that turned the accesses into flat_load / flat_store; the asm memory clobber below orders them)
  const int shift = CSHIFT > 0 ? CSHIFT : shift_rt;
  const int lvl = blockIdx.y;
  const int b = (BS_XCD_BINS && gridDim.x % 8 == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const uint32_t size = desc.size[lvl];
  const bool hashed = (desc.hashed_mask >> lvl) & 1u;
  const int nbins = (int)((size + (1u << shift) - 1) >> shift);
  if (!hashed || !is_pow2(size) || nbins > BS_MAX_BINS || nbins <= 1 || b >= nbins) return;
  const float gmax = lvl_max[lvl];
  if (!(gmax > 0.0f)) return;
  const uint32_t lo = (uint32_t)b << shift;
  if (nonfinite(gmax)) {
    if (threadIdx.x == 0) out[((size_t)desc.offset[lvl] + lo) * NV] = __builtin_nanf("");
    return;
  }
  const int seg = CSHIFT > 0 ? (1 << CSHIFT) : (1 << shift);
  const int n_ent = (int)min(1u << shift, size - lo);
  const int n_el = n_ent * NV;
  for (int i = threadIdx.x; i < seg * NV; i += blockDim.x) acc[i] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  owner_tag[wave][lane] = 0xFFFFFFFFu;
  __syncthreads();
  const float fxs = fx_scale(gmax * 16.5f, 30);
  // one record = NW words, kept as ONE register tuple from its load to its use (separate scalars made the register allocator copy
  // the words out of the load's destination right behind the load, i.e. wait for it there)
  typedef uint32_t RecVec __attribute__((ext_vector_type(NW)));
  auto add = [&](const RecVec rv) {  // as in bin_pass2_kernel
    const uint32_t w0 = rv[0];
    // (the halfs are taken out of the payload WORDS here, with shifts: read through a half_t pointer the words were split into
    // 16-bit pieces where they are loaded)
    auto hv_at = [&](int j) -> half_t { return __builtin_bit_cast(half_t, (unsigned short)(rv[1 + (j >> 1)] >> (16 * (j & 1)))); };
    const uint32_t local = w0 & ((1u << BS_KEY_BITS) - 1u), code = (w0 >> BS_KEY_BITS) & 15u;
    const bool single = code == BS_CODE_SINGLE;
    const float f1 = single ? 0.0f : (float)(w0 >> (BS_KEY_BITS + 4)) * (1.0f / BS_FX_ONE);
    const float s0 = (1.0f - f1) * fxs, s1 = f1 * fxs;
    const uint32_t other = local ^ ((2u << code) - 1u);
    unsigned long long* a0 = reinterpret_cast<unsigned long long*>(acc) + local;
    unsigned long long* a1 = reinterpret_cast<unsigned long long*>(acc) + other;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float v = h2f(hv_at(j));
      if (v != 0.0f) {
        const float2_t p = float2_t{s0, s1} * v;  // one v_pk_mul_f32
        atomicAdd(a0 + j * seg, (unsigned long long)(long long)fx_round(p[0]));
        if (!single) atomicAdd(a1 + j * seg, (unsigned long long)(long long)fx_round(p[1]));
      }
    }
  };
  const uint16_t* o0 = offs + ((uint64_t)lvl * (BS_MAX_BINS + 1) + b) * n_wg;
  const uint16_t* o1 = o0 + n_wg;
  constexpr uint32_t SLOT = (uint32_t)(BS_THREADS * NC) * NW;
  const uint32_t* lvl_bins = bins + (uint64_t)lvl * n_wg * SLOT;
  uint32_t stamp = 0u;
  const int n_batches = (n_wg + 63) >> 6;
  // the offsets of a wavefront's NEXT batch are fetched while the current one is walked
  int w_n = wave * 64 + lane;
  uint32_t s0_n = 0u, s1_n = 0u;
  if (wave < n_batches && w_n < n_wg) { s0_n = o0[w_n]; s1_n = o1[w_n]; }
  // Walk state (wave-uniform): the open batch of 64 tiles and the window position inside its concatenated runs.
  int batch = wave - n_waves, tile0 = 0, base = 0;
  uint32_t c = 0u, P0 = 0u, T = 0u, q0 = 0u;
  // fetch: the next window of 64 records (opening the next batches as needed) -- finds every position's owner and ISSUES the record
  // loads into (key, wd); nothing here waits for them.  false: no records left.
  auto fetch = [&](RecVec& rv, bool& ok) -> bool {
    while (q0 >= T) {  // (wave-uniform)
      batch += n_waves;
      if (batch >= n_batches) return false;
      const uint32_t s0 = s0_n;
      c = s1_n - s0_n;
      tile0 = batch * 64;
      {
        const int wn = (batch + n_waves) * 64 + lane;
        s0_n = s1_n = 0u;
        if (batch + n_waves < n_batches && wn < n_wg) { s0_n = o0[wn]; s1_n = o1[wn]; }
      }
      // inclusive prefix sum of the run lengths over the wavefront (DPP: row scan, then the row totals carried upwards)
      uint32_t inc = c;
#define L4D_ADD_DPP(ctrl, rmask) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, ctrl, rmask, 0xf, true)
      L4D_ADD_DPP(0x111, 0xf);  // row_shr:1 (bound_ctrl: lanes without a source add 0)
      L4D_ADD_DPP(0x112, 0xf);
      L4D_ADD_DPP(0x114, 0xf);
      L4D_ADD_DPP(0x118, 0xf);
      L4D_ADD_DPP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
      L4D_ADD_DPP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef L4D_ADD_DPP
      P0 = inc - c;                                                   // records of the batch in front of this tile's run
      T = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);         // records of the batch
      base = (int)s0 - (int)P0;                                       // record position q -> index in the tile's slot
      q0 = 0u;
    }
    ++stamp;
    // this tile's run enters the window [q0, q0 + 64) at position max(P0, q0) - q0, if it overlaps it at all
    if (c != 0u && P0 < q0 + 64u && P0 + c > q0) owner_tag[wave][P0 > q0 ? P0 - q0 : 0u] = (stamp << 8) | (uint32_t)lane;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t tv = owner_tag[wave][lane];
    int own = (tv >> 8) == stamp ? (int)(tv & 255u) : -1;
#define L4D_MAXI_DPP(ctrl, rmask) own = max(own, __builtin_amdgcn_update_dpp(-1, own, ctrl, rmask, 0xf, false))
    L4D_MAXI_DPP(0x111, 0xf);
    L4D_MAXI_DPP(0x112, 0xf);
    L4D_MAXI_DPP(0x114, 0xf);
    L4D_MAXI_DPP(0x118, 0xf);
    L4D_MAXI_DPP(0x142, 0xa);
    L4D_MAXI_DPP(0x143, 0xc);
#undef L4D_MAXI_DPP
    const uint32_t q = q0 + (uint32_t)lane;
    ok = q < T;  // (then own >= 0: position q0 lies inside a run, whose tile stamped position 0)
    const int ob = __shfl(base, ok ? own : lane, 64);
    // UNCONDITIONAL loads (lanes behind the batch's last record read the level's first record and ignore it): loads inside an
    // `if (ok)` made the compiler wait for them at the end of that block -- the merge with the default values is a copy
    const uint32_t* rec = ok ? lvl_bins + (uint64_t)(uint32_t)(tile0 + own) * SLOT + (uint32_t)(ob + (int)q) * NW : lvl_bins;
    __builtin_memcpy(&rv, rec, NW * sizeof(uint32_t));  // one global_load_dwordx2 / x3 (4-byte aligned; a 3-vector's size is 16)
    q0 += 64u;
    return true;
  };
  // Two windows in flight: the records of window n + 1 are requested before window n is accumulated (~100 VALU instructions and up
  // to 2 NV LDS atomics per record), two register sets in turn so that no copy waits for a load.  (One window at a time, every
  // wavefront sat out a full memory latency per 64 records: 1.86 ms against an issue floor of 1.0, profiles/r04_floor_table.md.)
  RecVec rec_a = {}, rec_b = {};
  bool ok_a, ok_b;
  // The empty asm is where a window's records are waited for: BEFORE the next window is requested, so that exactly one request is
  // outstanding at every wait (the compiler's s_waitcnt at a control-flow join is vmcnt(0): with two requests in flight it waited
  // for the newer one as well, right behind its issue); the request then has the whole accumulation of the previous window to land.
  bool more = fetch(rec_a, ok_a);
  while (more) {
    asm volatile("" : "+v"(rec_a));
    const bool more_b = fetch(rec_b, ok_b);
    if (ok_a) add(rec_a);
    if (!more_b) break;
    asm volatile("" : "+v"(rec_b));
    more = fetch(rec_a, ok_a);
    if (ok_b) add(rec_b);
  }
  __syncthreads();
  const double inv = (double)out_scale / (double)fxs;
  float* o = out + ((size_t)desc.offset[lvl] + lo) * NV;
  for (int i = threadIdx.x; i < n_el; i += blockDim.x) {
    const long long v = acc[(i % NV) * seg + i / NV];
    if (v != 0) o[i] += (float)((double)v * inv);
  }
}

// ---- host side ----------------------------------------------------------------------------------
// Entries per bin = 2^shift: the bin's int64 accumulators take 2^shift * NV * 8 bytes of LDS in pass 2 (64 KB -> two
// workgroups per CU, which is what hides the latency of the record walk), and more, smaller bins spread the pass-1
// histogram atomics.  L4D_BS_SHIFT4 / L4D_BS_SHIFT2 override (tuning).
static int bs_shift(int NV) {
  const char* e = getenv(NV == 4 ? "L4D_BS_SHIFT4" : NV == 2 ? "L4D_BS_SHIFT2" : "L4D_BS_SHIFT1");
  if (e && atoi(e) >= 9 && atoi(e) <= BS_KEY_BITS) return atoi(e);
  return NV == 4 ? 11 : NV == 2 ? 11 : 13;  // (NV = 2, the flow grid: 4,096-entry bins 1.01 ms in the flattened pass 2, 2,048: 0.89, 1,024: 0.93)
}

// L4D_BS_FLAT=0 selects the run-per-lane-group form of pass 2 (bin_pass2_kernel); default: the flattened walk
static bool bs_flat() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("L4D_BS_FLAT"); v = (e && atoi(e) == 0) ? 0 : 1; }
  return v != 0;
}

BsPlan bs_plan(const GridDesc& d, int n_dims, int NV, int64_t P) {
  BsPlan pl;
  pl.shift = bs_shift(NV);
  pl.rec_words = 1 + (NV + 1) / 2;
  pl.n_wg = ceil_div64(P, BS_THREADS);
  const int64_t rec_per_wg = (int64_t)BS_THREADS << n_dims;  // room for 2^D records per lane (only those that exist are written)
  pl.off_max = 0;
  pl.off_offs = 256;
  pl.off_bins = (pl.off_offs + (int64_t)d.n_levels * pl.n_wg * (BS_MAX_BINS + 1) * 2 + 255) / 256 * 256;
  pl.bytes = pl.off_bins + (int64_t)d.n_levels * pl.n_wg * rec_per_wg * pl.rec_words * 4;
  return pl;
}

int bs_scatter(const GridDesc& desc, int n_dims, int NV, const float* x, int64_t P, int x_stride, const int* cols, const half_t* g,
               int g_stride, int g_col, float pre_scale, float* out, float out_scale, void* workspace, hipStream_t stream) {
  if (P == 0) return 0;
  const BsPlan pl = bs_plan(desc, n_dims, NV, P);
  char* ws = (char*)workspace;
  float* lvl_max = (float*)(ws + pl.off_max);
  uint16_t* offs = (uint16_t*)(ws + pl.off_offs);
  uint32_t* bins = (uint32_t*)(ws + pl.off_bins);
  l4d_fill_async(ws, 0u, 256, stream);
  BsCols c;
  for (int d = 0; d < 3; ++d) c.c[d] = d < n_dims ? cols[d] : 0;
  if ((int64_t)desc.n_levels * (BS_MAX_BINS + 1) * pl.n_wg >= ((int64_t)1 << 32)) { l4d_set_error(1, "bs_scatter: too many points for one launch"); return 1; }
  for (int l = 0; l < desc.n_levels; ++l) {  // the pair records carry their code in key bits 24..27: BINNED levels only (hashed,
    // power-of-two table, <= BS_MAX_BINS bins); larger tables never reach the binned path (atomic fallback) and need no limit
    const int64_t nb = ((int64_t)desc.size[l] + (1 << pl.shift) - 1) >> pl.shift;
    const bool binned = ((desc.hashed_mask >> l) & 1u) && is_pow2(desc.size[l]) && nb <= BS_MAX_BINS && nb > 1;
    if (binned && desc.size[l] > (1u << 24)) { l4d_set_error(1, "bs_scatter: more than 2^24 entries in a binned level"); return 1; }
  }
  int max_bins = 1;
  for (int l = 0; l < desc.n_levels; ++l) max_bins = std::max<int>(max_bins, (int)(((int64_t)desc.size[l] + (1 << pl.shift) - 1) >> pl.shift));
  max_bins = std::min(max_bins, BS_MAX_BINS);
  dim3 grid1((unsigned)xcd_grid(pl.n_wg));
  dim3 grid2(max_bins, desc.n_levels);
  const int lds2 = (1 << pl.shift) * NV * 8;
#define BS_LAUNCH(D, V)                                                                                                      \
  {                                                                                                                          \
    L4D_LAUNCH((bin_pass1_kernel<D, V>), grid1, dim3(BS_THREADS), 0, stream, desc, x, P, x_stride, c, g, g_stride,   \
                       g_col, pre_scale, pl.shift, (int64_t)pl.n_wg, offs, bins, lvl_max, out, out_scale);                                     \
    constexpr int DEF = V == 4 ? 11 : V == 2 ? 11 : 13;  /* bs_shift()'s defaults: compile-time bin size */                       \
    if (pl.shift == DEF && bs_flat()) {                                                                                      \
      (void)hipFuncSetAttribute((const void*)bin_pass2_flat_kernel<D, V, DEF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);  \
      L4D_LAUNCH((bin_pass2_flat_kernel<D, V, DEF>), grid2, dim3(1024), lds2, stream, desc, pl.shift, (int)pl.n_wg, P, offs, bins, \
                 lvl_max, out, out_scale);                                                                                   \
    } else if (bs_flat()) { /* a bin size other than the compiled-in one (L4D_BS_SHIFT*: tuning) */                          \
      (void)hipFuncSetAttribute((const void*)bin_pass2_flat_kernel<D, V, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);    \
      L4D_LAUNCH((bin_pass2_flat_kernel<D, V, 0>), grid2, dim3(1024), lds2, stream, desc, pl.shift, (int)pl.n_wg, P, offs, bins,   \
                 lvl_max, out, out_scale);                                                                                   \
    } else if (pl.shift == DEF) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)bin_pass2_kernel<D, V, DEF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);       \
      L4D_LAUNCH((bin_pass2_kernel<D, V, DEF>), grid2, dim3(1024), lds2, stream, desc, pl.shift, (int)pl.n_wg, P, offs, bins,  \
                 lvl_max, out, out_scale);                                                                                   \
    } else {                                                                                                                 \
      (void)hipFuncSetAttribute((const void*)bin_pass2_kernel<D, V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);            \
      L4D_LAUNCH((bin_pass2_kernel<D, V>), grid2, dim3(1024), lds2, stream, desc, pl.shift, (int)pl.n_wg, P, offs, bins,       \
                 lvl_max, out, out_scale);                                                                                   \
    }                                                                                                                        \
  }
  if (n_dims == 3 && NV == 4) BS_LAUNCH(3, 4)
  else if (n_dims == 3 && NV == 2) BS_LAUNCH(3, 2)
  else if (n_dims == 3 && NV == 1) BS_LAUNCH(3, 1)
  else if (n_dims == 2 && NV == 4) BS_LAUNCH(2, 4)
  else if (n_dims == 2 && NV == 2) BS_LAUNCH(2, 2)
  else if (n_dims == 2 && NV == 1) BS_LAUNCH(2, 1)
  else { l4d_set_error(1, "bs_scatter: unsupported dims / payload width"); return 1; }
#undef BS_LAUNCH
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { l4d_set_error((int)le, "bs_scatter"); return (int)le; }
  return 0;
}
