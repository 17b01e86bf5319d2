// Sorted gradient scatter for the large 3-D hash tables (static grid 2^19 x 4, flow grid 2^18 x 8) on gfx950.
//
// Scattered global fp32 atomics sustain ~20 G lane-ops/s on MI355X (profiles/r01_ubench_global_atomics.txt);
// a 16,384-ray step needs 3.2 G of them for the static grid alone (148 ms measured).  Tables of 8 MB per level do
// not fit LDS, so contributions are first *binned* by table segment and then reduced segment by segment in LDS:
//
//   pass 1  (one thread per (sample, level)): the 2^D corner contributions {entry, w * g[0..NV)} (fp16 payload) are
//           partitioned inside the workgroup by bin = entry >> shift (LDS histogram + ranks), staged sorted by bin in LDS,
//           and appended to the BIN-MAJOR record lists in HBM: one returning global atomic per (workgroup, level, bin)
//           reserves the run's place in list [level][bin][XCD of the workgroup] (round 6; rounds 1-5 kept every workgroup's
//           records in its own slot, sorted by bin, plus a table of bin offsets -- pass 2 then read ~100-byte runs at 4-byte
//           alignment out of 24,576 slots: 9.15 GB of 128-byte line fills for 4.8 GB of records at an L2 hit rate of 17 %).
//           Coarse levels first merge runs of samples in one cell along the ray, over the whole wavefront (DPP: wave_dev.h wave_scan);
//           the level maximum pass 1 reports covers those totals (pass 2 scales its fixed point by it).
//   pass 2  (one workgroup per (level, bin)): STREAMS its bin's eight lists -- contiguous, every line read once, whole --
//           accumulates in LDS as int64 fixed point (ds_add_u64: 2.9 T lane-ops/s, exact and order independent), and
//           adds the segment to the fp32 gradient table with plain stores -- each segment has exactly one owner.
//
// LISTS AND OVERFLOW.  A list holds `cap` records = 1.25 x its expected share (samples x 2^(D-1) pair records / (bins x 8)) +
// 8 sigma + 64: a hashed table spreads the records evenly, so this is never reached by data that looks like a scene.  It is
// still only a capacity: records that do not fit (clustered or repeated points, pairs that all straddle bins) go, tagged with
// their bin, to the level's OVERFLOW list, which has room for every record the level can produce (samples x 2^D); the owner
// of a bin that overflowed (a per-bin counter says so) scans that list after its own.  Where a record lands depends on the
// arrival of atomics; WHAT is summed does not, and the sums are integers: results stay bit-reproducible in every case
// (tests/test_gpu_properties.py: random points, and all points in one cell).  Lists are per XCD (workgroup b runs on XCD
// b % 8: speed only) so that the tail line of a list is completed inside one L2 before it is written back, and so that
// eight counters, not one, take a bin's reservations.
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
#ifndef BS_SCAN_MAX_HEADS
#define BS_SCAN_MAX_HEADS 32  // merged-run form of a binned level up to this many runs per wavefront (<= 32: at most NC / 2 records per lane)
#endif
#define BS_MIN_WAVES 4  // pass 1: wavefronts per SIMD the register allocation must leave room for (128 registers)

// key + packed halfs.  Records are stored whole (8 or 12 bytes).  ``split`` stores 12-byte records as a dword key stream + an 8-byte
// payload stream per level, same record index in both (naturally aligned stores and loads); measured slower than whole records at
// 12-byte alignment (pass 1 2.04 -> 2.16 ms, session s5 of round 6: two store instructions per record cost more than their alignment
// saves), kept as a compile-time option of the layout.
template <int NV>
struct RecWords {
  static constexpr int n = 1 + (NV + 1) / 2;
  static constexpr bool split = false;
};

template <int NV>
__device__ __forceinline__ void pack_payload(const float v[NV], uint32_t out[(NV + 1) / 2]) {
#pragma unroll
  for (int q = 0; q < (NV + 1) / 2; ++q) {
    const half2_t h = {f2h_grad(v[2 * q]), 2 * q + 1 < NV ? f2h_grad(v[2 * q + 1]) : (half_t)0.0f};
    out[q] = __builtin_bit_cast(uint32_t, h);
  }
}

// -DBS_PHASE_CLOCK (tools/build_abl.sh, tools/bs_phase.py): cycles (s_memtime) that the wavefronts of pass 1 spend in every phase of a
// level, summed over the launch -- [0: wavefront 0 (the scanning one), 1: the others][phase]; not compiled into the shipped library
#ifdef BS_PHASE_CLOCK
__device__ unsigned long long bs_phase_clk[2][18];
// (read out and reset by a kernel into DEVICE memory the caller owns -- the library issues no memcpy / memset calls: graph safety)
__global__ void bs_phase_clk_read_kernel(unsigned long long* __restrict__ out, int reset) {
  const int i = threadIdx.x;
  if (i >= 36) return;
  unsigned long long* src = &bs_phase_clk[0][0];
  if (out) out[i] = src[i];
  if (reset) src[i] = 0ull;
}
extern "C" int l4d_debug_bs_phase_clk(unsigned long long* out_dev, int reset, void* stream) {
  bs_phase_clk_read_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out_dev, reset);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
// (sampled: wavefronts 1 -- one of the four that reserve -- 4 and 7 of every 16th workgroup -- s_memtime from every wavefront made the kernel ten times slower)
#define BS_CLK_DECL const bool clk_on = (blockIdx.x & 15) == 0 && (threadIdx.x >> 6) % 3 == 1; uint32_t clk_acc[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint32_t clk_last = clk_on ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
#define BS_CLK(i) if (clk_on) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); clk_acc[i] += now_ - clk_last; clk_last = now_; }
#define BS_CNT(i) if (clk_on) clk_acc[i] += 1u;
#else
#define BS_CLK_DECL
#define BS_CLK(i)
#define BS_CNT(i)
#endif

template <int D, int NV>
__global__ void __launch_bounds__(BS_THREADS, BS_MIN_WAVES) bin_pass1_kernel(GridDesc desc, const float* __restrict__ x, int64_t P, int x_stride,
                                                              BsCols cols, const half_t* __restrict__ g, int g_stride, int g_col,
                                                              float pre_scale, int shift, int64_t n_wg, BsLayout lay,
                                                              uint32_t* __restrict__ cur, uint32_t* __restrict__ ovf_cur, uint32_t* __restrict__ ovf_cnt,
                                                              uint32_t* __restrict__ lists, uint32_t* __restrict__ ovf,
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
  __shared__ uint8_t rbin[CAP];  // bin of every staged record (the copy-out's destination depends on it)
  // what the copy-out of a level needs, double-buffered by the parity of the binned level (the copy-out runs under the NEXT level's
  // bin scan, which produces that level's values): per bin ONE 16-byte entry {i0, lim, -, odel} -- record r of the stage (bin b) is
  // record i0 + r of the level's lists if r < lim, else record r + odel of the level's overflow list; the number of records staged;
  // and whether any list of the level was full (then, and only then, the copy-out runs its second loop)
  __shared__ __attribute__((aligned(16))) uint4 bdst[2][BS_MAX_BINS];
  __shared__ uint32_t total_s[2], ovf_flag[2];
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
  // ranked.
  constexpr int GW = 8;  // (round 6, first layout: 8 dwords left no room for the list reservations at 126 registers, 2.00 -> 2.13 ms, session s18; with the
                         // wave-wide merge the kernel holds 116: 2.04 -> 2.01 ms and 3 GB less over the fabric, session s28)
  const bool g_in_regs = n_lv * NV <= 32 && (g_stride * 2) % 16 == 0 && (g_col * 2) % 16 == 0;  // block-uniform
  typedef uint32_t GwVec __attribute__((ext_vector_type(GW)));  // a vector, so that a uniform index becomes relative VGPR addressing
  GwVec gw;
  // (g_in_regs only.  The loads are unconditional -- a piece behind the last level re-reads piece 0 and is never picked: a load
  // under a condition merges with its default value in register copies, and the copies wait for the load where it is issued)
  auto gw_load = [&](int chunk) {  // dwords GW chunk .. GW chunk + GW - 1 of the row's gradient columns
#pragma unroll
    for (int q = 0; q < GW / 4; ++q) {
      const int piece = (chunk * (GW / 4) + q) * 8 < n_lv * NV ? chunk * (GW / 4) + q : 0;  // block-uniform
      const uint4 u = *reinterpret_cast<const uint4*>(grow + piece * 8);
      gw[4 * q + 0] = u.x; gw[4 * q + 1] = u.y; gw[4 * q + 2] = u.z; gw[4 * q + 3] = u.w;
    }
  };
#pragma unroll
  for (int q = 0; q < GW; ++q) gw[q] = 0u;
  if (g_in_regs) gw_load(0);
  auto gw_pick = [&](int k) -> uint32_t { return gw[k & (GW - 1)]; };
  half_t gnext[NV];
  for (int i = threadIdx.x; i < BS_MAX_BINS; i += BS_THREADS) hist[i] = 0;
  if (threadIdx.x < L4D_MAX_LEVELS) lmax_s[threadIdx.x] = 0u;
  BS_CLK_DECL
  __syncthreads();
  BS_CLK(0)  // head: coordinates, first gradient piece, first barrier
  // The level loop exists twice (a generic lambda over GREG = "gradient dwords in the register window"): the loads of the other
  // shapes' path, in ONE loop with the common path, made the compiler wait for every outstanding memory operation at the join --
  // the gradient dwords just requested and the previous level's copy-out stores included -- on the common path as well.
  // COPY-OUT, UNDER THE NEXT LEVEL'S BIN SCAN.  A level's staged records (sorted by bin) leave LDS between the first and the second
  // barrier of the next binned level -- while wavefront 0 scans that level's bin totals the other seven copy -- or behind the loop.
  // Consecutive records of a run are consecutive in their list, so a wavefront's store covers ~8 runs = 8-16 lines.  The reservation
  // behind bdst is a returning global atomic that the bin's thread issues behind the level's own scan and consumes one ranking phase
  // later (finish_reserve): its round trip never stands in anybody's way.  Nothing touches `stage`, rbin or this parity's bdst /
  // total_s before the next level's second barrier, which every wavefront reaches only behind its share of the copy.
  // The loop is software-pipelined by hand: a record's bin (one byte) is read an iteration ahead, so that an iteration waits once --
  // for its bdst entry and its record, requested together -- not twice in a row.
  const int list_k = (int)(blockIdx.x & (BS_LISTS - 1));  // workgroup b runs on XCD b % 8 (observed; speed only)
  int copy_lvl = -1, nb = 0;  // block-uniform: level whose records are staged; binned levels seen so far (parity = nb & 1)
  auto copy_out = [&](int first, int step) {
    const int par = (nb - 1) & 1;
    const uint32_t total = total_s[par];
    // the level's record region (split records: key stream, then payload stream; RecWords)
    typedef __attribute__((address_space(1))) uint32_t GlobalU32;
    typedef __attribute__((address_space(1))) char GlobalByte;
    const uint64_t key_base = reinterpret_cast<uint64_t>(lists + lay.list_base[copy_lvl] * NW);
    const uint64_t lvl_recs = (uint64_t)(desc.size[copy_lvl] >> shift) * BS_LISTS * lay.cap[copy_lvl];
    const uint64_t pay_base = key_base + lvl_recs * 4u;
    const bool near = lvl_recs * (NW * 4u) < (1ull << 32);  // block-uniform: 32-bit byte offsets from a scalar base reach every record
    // HOT LOOP: records that fit their list (all of them, unless ovf_flag says otherwise); one multiply for the address
    {
      uint32_t r = (uint32_t)first;
      uint32_t b_next = rbin[r < CAP ? r : 0u];
      while (r < total) {
        const uint32_t b = b_next;
        const uint32_t rn = r + (uint32_t)step;
        b_next = rbin[rn < CAP ? rn : 0u];
        uint2 e = *reinterpret_cast<const uint2*>(&bdst[par][b]);  // {i0, lim}
        uint32_t w[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) w[q] = stage[r * NW + q];
        // (the entry and the record requested together, ONE wait)
        if (NW == 3) asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(w[0]), "+v"(w[1]), "+v"(w[NW - 1]));
        else asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(w[0]), "+v"(w[NW - 1]));
        if (r < e.y) {
          const uint32_t rec = e.x + r;  // record index inside the level's region (mod 2^32: e.x is relative to stage record 0)
          if (RecWords<NV>::split) {
            *(GlobalU32*)(key_base + (uint64_t)rec * 4u) = w[0];
            __builtin_memcpy((GlobalU32*)(pay_base + (uint64_t)rec * ((NW - 1) * 4u)), w + 1, (NW - 1) * sizeof(uint32_t));
          } else {  // ONE store of the whole record (a vector type of 4-byte alignment: global_store_dwordx2 / x3)
            typedef uint32_t RecVecA __attribute__((ext_vector_type(NW), aligned(4)));
            typedef __attribute__((address_space(1))) RecVecA GlobalRec;
            RecVecA rv;
#pragma unroll
            for (int q = 0; q < NW; ++q) rv[q] = w[q];
            if (near) *(GlobalRec*)((GlobalByte*)key_base + rec * (uint32_t)(NW * 4u)) = rv;
            else *(GlobalRec*)(key_base + (uint64_t)rec * (NW * 4u)) = rv;
          }
        }
        r = rn;
      }
    }
    if (ovf_flag[par]) {  // (block-uniform, rare) records of lists that are full: LISTS AND OVERFLOW
      const uint32_t ovf_cap = lay.ovf_cap[copy_lvl];
      uint32_t* ov = ovf + lay.ovf_base[copy_lvl] * (NW + 1);
      for (uint32_t r = (uint32_t)first; r < total; r += (uint32_t)step) {
        const uint32_t b = rbin[r];
        const uint4 e = bdst[par][b];
        const uint32_t op = r + e.w;
        if (r >= e.y && op < ovf_cap) {
          uint32_t* d = ov + (uint64_t)op * (NW + 1);
          d[0] = b;
#pragma unroll
          for (int q = 0; q < NW; ++q) d[1 + q] = stage[r * NW + q];
        }
      }
    }
  };
  // threads 0 .. 255 own one bin each: the reservation of the staged level's run for that bin has returned by now; turn it into the
  // bin's bdst entry
  constexpr int BPL = BS_MAX_BINS / 64;
  uint32_t resv = 0u;
  auto finish_reserve = [&]() {
    const int par = (nb - 1) & 1;
    const uint32_t cap = lay.cap[copy_lvl];
    uint32_t b = threadIdx.x;
    asm volatile("" : "+v"(b));  // (opaque per level: hoisted out of the level loop, the LDS and cursor addresses of every use of b cost a register each -- spilled)
    const uint32_t o0 = boff[b], cnt = boff[b + 1] - o0;
    const uint32_t fit = resv >= cap ? 0u : min(cnt, cap - resv);  // records of the run that fit the list
    uint32_t odel = 0u;
    if (fit < cnt) {  // (never with data that looks like a scene)
      const uint32_t n_ovf = cnt - fit;
      const uint32_t op = atomicAdd(ovf_cur + copy_lvl, n_ovf);
      atomicAdd(ovf_cnt + copy_lvl * BS_MAX_BINS + b, n_ovf);
      odel = op - (o0 + fit);
      ovf_flag[par] = 1u;
    }
    // {record index of stage record 0 if the run started there (mod 2^32), first stage index that does not fit, -, overflow delta}
    bdst[par][b] = uint4{(b * BS_LISTS + (uint32_t)list_k) * cap + resv - o0, o0 + fit, 0u, odel};
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

  // THE LEVEL'S GRADIENT VALUES ARE TAKEN LATE: behind the cell location, the corner hashes and the weights, none of which needs
  // them.  Taking them is where the compiler waits for EVERY outstanding memory operation of the wavefront (its counter wait in a
  // loop is vmcnt(0)) -- the previous level's copy-out stores and, in the first four wavefronts, the returning atomic that reserved
  // the previous level's runs (issued two barriers back).  At the head of the level that wait stood ~2 us of atomic round trip in
  // front of everything (session s4 of round 6: 0.28 ms of pass 1); here the hashing has covered it.
  float gv[NV];
  bool any = false;
  float amax = 0.0f;
  auto take_gradient = [&]() {
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
      // first dword of the next level (NV = 1: two levels per dword); at a 16-byte boundary the held dwords are all consumed
      const int k_next = NV == 1 ? (lvl + 1) >> 1 : (lvl + 1) * (NV / 2);
      if (lvl + 1 < n_lv && k_next % GW == 0 && (NV > 1 || ((lvl + 1) & 1) == 0)) gw_load(k_next / GW);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      gv[j] = valid ? h2f(gnext[j]) * pre_scale : 0.0f;
      any |= gv[j] != 0.0f;
      amax = amax_nf(amax, gv[j]);
    }
  };
  // (No barrier here: the histogram was left zeroed by the scan of the previous binned level, and this level's staging writes and
  // bin offsets come behind the barriers below, which every wavefront reaches only behind its share of the previous level's copy-out.)

  BS_CLK(11)  // (behind the second barrier:) reservation issued, records staged
  Cell<D> c = locate<D>(xin, desc.scale[lvl]);
  uint32_t keys[NC];
  float vals[NC][NV];
  bool emit[NC];
  uint32_t pos[NC];
  // Consecutive lanes are consecutive samples of a ray: on coarse levels several of them sit in one cell and hit the same
  // 2^D entries.  Such runs are merged over the whole wavefront with the DPP scan of wave_dev.h -- one record (binned levels) or
  // one atomic (dense levels) per run, corner and value.  (Round 6: the dense levels used the shuffle reduction wave_run_reduce,
  // 6 ds_bpermute round trips per value and corner -- half of the 14 k cycles a workgroup spent on the flow grid's dense level,
  // tools/bs_phase.py; merging inside 16-lane rows only tripled their atomics, which all rays aim at the same few cells around
  // the sensor: pass 1 2x slower.)
  int n_heads = 64;
  WaveRuns runs;
  bool use_scan = false;
  {
    bool same = true;  // same cell as the previous lane (lane 0 never is: old = ~cell, bound_ctrl off)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp((int)~c.cell[d], (int)c.cell[d], 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
      same &= pv == c.cell[d];
    }
    runs = wave_runs(__ballot(!same), &n_heads);  // (lanes without a gradient carry zeros: harmless in any run)
    use_scan = binned && n_heads <= BS_SCAN_MAX_HEADS;  // wave-uniform: merged runs emit single records, NC per run
  }
  const bool pairs = binned && !use_scan;  // wave-uniform
  uint32_t fxq = 0u;  // keys[] carry the record's tz code in bits 24..27 from here on (entries per level <= 2^24: checked by the host side)
  bool wave_any;
  if (pairs) {
    // one record per x-neighbour pair (corners 2q, 2q + 1): slots 0 .. NC/2 - 1 are used, the rest stay silent
    fxq = (uint32_t)fx_round(c.frac[0] * BS_FX_ONE);
    const float fx = (float)fxq * (1.0f / BS_FX_ONE);
    float wyz[NC / 2];  // weight without the x factor
    bool paired[NC / 2];
#pragma unroll
    for (int q = 0; q < NC / 2; ++q) {
      uint32_t g0[D], g1[D];
      (void)corner<D>(c, 2 * q, g0);
      (void)corner<D>(c, 2 * q + 1, g1);
      wyz[q] = 1.0f;
#pragma unroll
      for (int d = 1; d < D; ++d) wyz[q] *= ((2 * q) >> d) & 1 ? c.frac[d] : 1.0f - c.frac[d];
      const uint32_t k0 = grid_index_fast<D>(g0, size - 1u), k1 = grid_index_fast<D>(g1, size - 1u);
      const uint32_t m = k0 ^ k1;
      paired[q] = (m & (m + 1u)) == 0u && m != 0u && (m >> shift) == 0u && __popc(m) <= (int)BS_CODE_SINGLE;
      keys[q] = k0 | ((paired[q] ? (uint32_t)(__popc(m) - 1) : BS_CODE_SINGLE) << 24);
      keys[q + NC / 2] = k1 | (BS_CODE_SINGLE << 24);
    }
    BS_CLK(1)  // pair form: cell, hashes, weights
    take_gradient();
#ifdef BS_PHASE_CLOCK
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(gv[j]));  // (the values exist before the clock is read)
#endif
    BS_CLK(2)  // the level's gradient values (the wait for outstanding memory operations)
    wave_any = __any(any);
#pragma unroll
    for (int q = 0; q < NC / 2; ++q) {
      float* v = vals[q];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = wyz[q] * gv[j];
      bool nz = false;
#pragma unroll
      for (int j = 0; j < NV; ++j) nz |= v[j] != 0.0f;
      emit[q] = any && nz;
      // straddles two bins (2^-shift of the pairs): two single records, the neighbour's in the otherwise unused slot q + NC/2
      const bool split = emit[q] && !paired[q];
      emit[q + NC / 2] = split;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        vals[q + NC / 2][j] = v[j] * fx;
        if (split) v[j] *= 1.0f - fx;
      }
    }
  } else {
    float wk[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      uint32_t gg[D];
      wk[k] = corner<D>(c, k, gg);
      keys[k] = binned ? grid_index_fast<D>(gg, size - 1u) : grid_index<D>(gg, desc.res[lvl], size, hashed);
    }
#ifdef BS_PHASE_CLOCK
#pragma unroll
    for (int k = 0; k < NC; ++k) asm volatile("" : "+v"(keys[k]), "+v"(wk[k]));
#endif
    BS_CLK(12)  // merged-run / dense form: cell, run flags, hashes, weights
    take_gradient();
#ifdef BS_PHASE_CLOCK
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(gv[j]));
#endif
    BS_CLK(13)  // ... gradient values
    wave_any = __any(any);
#pragma unroll
  for (int k = 0; k < NC; ++k) {
#pragma unroll
    for (int j = 0; j < NV; ++j) vals[k][j] = wk[k] * gv[j];
    if (binned) {
      emit[k] = any;
      if (use_scan) {
        wave_scan<NV>(runs, vals[k]);
        emit[k] = runs.tail;
        // a merged run's total (up to 64 contributions) is what the record carries: the level's maximum -- pass 2's fixed-point bound --
        // has to cover it, and a total beyond the fp16 range must not become a finite record (checked once, below)
#pragma unroll
        for (int j = 0; j < NV; ++j) amax = fmaxf(amax, fabsf(vals[k][j]));
      }
    } else {
      emit[k] = false;
      if (wave_any) {  // (wave-uniform)
        wave_scan<NV>(runs, vals[k]);
        emit[k] = runs.tail;
      }
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
#ifndef BS_DENSE_NOATOMIC
          if (vals[k][j] != 0.0f) atomicAdd(o + (size_t)keys[k] * NV + j, vals[k][j] * out_scale);
#else
          if (vals[k][j] == 12345.0f) atomicAdd(o + (size_t)keys[k] * NV + j, vals[k][j] * out_scale);
#endif
      }
    BS_CLK(8)  // a dense level: scans, atomics
    BS_CNT(16)
    continue;  // block-uniform: next level
  }
  if (!pairs) { BS_CLK(10) BS_CNT(15) } else { BS_CNT(14) }  // merged-run form of a binned level: values, scans
  // (65520 = where round-to-nearest-even turns into the fp16 infinity; an upstream inf / nan is in amax already -- amax_nf.  A partial
  // sum of a run may trip this with a total that fits: a skipped step and a halved loss scale, what an overflow costs anyway)
  if (use_scan && amax >= 65520.0f) amax = __builtin_inff();
  amax = wave_max(amax);
  if (lane == 0 && amax > 0.0f) atomicMax(&lmax_s[lvl], __float_as_uint(amax));  // (ds_max_u32, nothing returned: no wait)

  // In pair mode the upper half of the slots only holds the second halves of pairs that straddle two bins (one pair in 2^shift):
  // a wavefront without one skips their ranking and staging code altogether (wave-uniform branch instead of exec-masked no-ops).
  bool upper = true;
  if (pairs) {
    bool up = false;
#pragma unroll
    for (int k = NC / 2; k < NC; ++k) up |= emit[k];
    upper = __any(up);
  }
  // rank inside the workgroup
#pragma unroll
  for (int k = 0; k < NC / 2; ++k) pos[k] = emit[k] ? atomicAdd(&hist[(keys[k] & 0xFFFFFFu) >> shift], 1u) : 0u;
  if (upper) {
#pragma unroll
    for (int k = NC / 2; k < NC; ++k) pos[k] = emit[k] ? atomicAdd(&hist[(keys[k] & 0xFFFFFFu) >> shift], 1u) : 0u;
  }
  // (the reservations of the level that is still staged have had this level's head and ranking phase to return)
  BS_CLK(3)  // records, ranks (returned)
  if (threadIdx.x < BS_MAX_BINS && copy_lvl >= 0) finish_reserve();
  BS_CLK(4)  // records, ranks, the previous level's reservation consumed (the clock is read where the barrier waits for the LDS anyway)
  __syncthreads();
  BS_CLK(5)  // first barrier
  const int par = nb & 1;
  if (threadIdx.x < 64) {  // exclusive scan of the bin totals by one wave: BPL consecutive bins per lane
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
    uint32_t excl = inc - sum;
#pragma unroll
    for (int q = 0; q < BPL; ++q) {
      const int b = lane * BPL + q;
      boff[b] = excl;
      excl += c[q];
    }
    if (lane == 63) {
      total_s[par] = inc;
      ovf_flag[par] = 0u;
      boff[BS_MAX_BINS] = inc;
    }
  } else if (copy_lvl >= 0) {
    copy_out((int)threadIdx.x - 64, BS_THREADS - 64);  // the previous binned level's records, under the scan
  }
  BS_CLK(6)  // scan (wavefront 0) / copy-out (the others)
  __syncthreads();
  BS_CLK(7)  // second barrier
  // this level's reservations: ONE returning atomic per bin on the list cursor [level][list][bin], issued by the bin's thread (the
  // first four wavefronts; 64 lanes = 64 consecutive counters = two lines).  Unconditional (a bin without records adds 0): a
  // conditional one is waited for where it is issued.  Consumed by finish_reserve() in front of the NEXT level's first barrier.
  if (threadIdx.x < BS_MAX_BINS) {
    uint32_t b = threadIdx.x;
    asm volatile("" : "+v"(b));  // (as above: scalar base + 32-bit lane offset, formed here)
    resv = atomicAdd(cur + ((uint32_t)lvl * BS_LISTS + (uint32_t)list_k) * BS_MAX_BINS + b, boff[b + 1] - boff[b]);
  }
  // The bins' offsets for ALL slots are requested first, unconditionally (a slot without a record reads its key's bin all the same):
  // read inside the slot's own branch, each offset was an LDS round trip of its own -- eight in a row per level, a fifth of the
  // kernel's time (tools/bs_phase.py) -- and every wait for one also waited for the previous slot's stores.
  uint32_t sbase[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) sbase[k] = (k < NC / 2 || upper) ? boff[(keys[k] & 0xFFFFFFu) >> shift] : 0u;  // (wave-uniform condition)
#pragma unroll
  for (int k = 0; k < NC; ++k) asm volatile("" : "+v"(sbase[k]));  // (all of them in flight before the first is used)
  auto stage_slot = [&](int k) {
    if (emit[k]) {
      const uint32_t b = (keys[k] & 0xFFFFFFu) >> shift;
      const uint32_t r = sbase[k] + pos[k];
      const uint32_t code = keys[k] >> 24;
      rbin[r] = (uint8_t)b;
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
  copy_lvl = lvl;  // the staged records leave under the next binned level's scan (or behind the loop)
  ++nb;
  }  // levels
  };
  if (g_in_regs) levels(std::true_type{});
  else levels(std::false_type{});
  if (copy_lvl >= 0) {  // the last binned level's records
    if (threadIdx.x < BS_MAX_BINS) finish_reserve();
    __syncthreads();
    copy_out((int)threadIdx.x, BS_THREADS);
  }
  __syncthreads();
  BS_CLK(9)  // tail: the last level's copy-out
#ifdef BS_PHASE_CLOCK
  if (lane == 0 && clk_on) {
#pragma unroll
    for (int i = 0; i < 18; ++i) atomicAdd(&bs_phase_clk[threadIdx.x < 256 ? 0 : 1][i], (unsigned long long)clk_acc[i]);
  }
#endif
  if ((int)threadIdx.x < n_lv) {
    const uint32_t m = lmax_s[threadIdx.x];
    if (m != 0u) atomic_max_nonneg(lvl_max + threadIdx.x, __uint_as_float(m));
  }
}

// ---- pass 2 -----------------------------------------------------------------------------------------------------------------
// One workgroup per (level, bin).  Its records are eight contiguous lists (one per XCD, pass 1 above): every wavefront takes 64
// consecutive records of a list at a time -- one 8- or 12-byte load per lane, whole lines, each read once -- and adds them to the
// bin's [value][entry] int64 accumulators in LDS; the records of a window are requested before the previous window is accumulated.
// (Rounds 3-5 walked every pass-1 workgroup's run for the bin instead: a table of bin offsets, and per 64 records a prefix scan
// over 64 run lengths, owner stamps in LDS, a max scan and a ds_bpermute to find out whose record a lane holds.)
// CSHIFT > 0: the bin size is a compile-time constant, so the accumulator addresses are ONE register (entry * 8) plus immediate
// offsets (j * 2^CSHIFT * 8 <= 48 KB fits the DS offset field).  CSHIFT = 0: run-time.
template <int D, int NV, int CSHIFT = 0>
__global__ void __launch_bounds__(1024) bin_reduce_kernel(GridDesc desc, int shift_rt, BsLayout lay, const uint32_t* __restrict__ cur,
                                                        const uint32_t* __restrict__ ovf_cur, const uint32_t* __restrict__ ovf_cnt,
                                                        const uint32_t* __restrict__ lists, const uint32_t* __restrict__ ovf,
                                                        const float* __restrict__ lvl_max, float* __restrict__ out, float out_scale) {
  constexpr int NW = RecWords<NV>::n;
  extern __shared__ long long acc[];
  const int shift = CSHIFT > 0 ? CSHIFT : shift_rt;
  const int lvl = blockIdx.y, b = blockIdx.x;
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
  // the eight list lengths (block-uniform: scalar loads) and their 64-aligned prefix: window w of the concatenation lies in ONE list
  const uint32_t cap = lay.cap[lvl];
  // (eight named scalars each, not arrays: the compiler turns a select chain over array elements into an indexed load from scratch)
  static_assert(BS_LISTS == 8, "the list walk below is written out for eight lists");
#define BS_LEN(k) const uint32_t n##k = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(cur[((uint32_t)lvl * BS_LISTS + k) * BS_MAX_BINS + b], cap));
  BS_LEN(0) BS_LEN(1) BS_LEN(2) BS_LEN(3) BS_LEN(4) BS_LEN(5) BS_LEN(6) BS_LEN(7)
#undef BS_LEN
#define BS_R64(n) (((n) + 63u) & ~63u)
  const uint32_t o1 = BS_R64(n0), o2 = o1 + BS_R64(n1), o3 = o2 + BS_R64(n2), o4 = o3 + BS_R64(n3), o5 = o4 + BS_R64(n4), o6 = o5 + BS_R64(n5),
                 o7 = o6 + BS_R64(n6), v_end = o7 + BS_R64(n7);
#undef BS_R64
  const uint32_t n_ovf_mine = ovf_cnt[lvl * BS_MAX_BINS + b];
  for (int i = threadIdx.x; i < seg * NV; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  // Fixed point: every contribution is |v| <= gmax, scaled to 30 bits and converted with ONE v_cvt_i32_f32 (a float -> int64
  // conversion is a dozen instructions on this ISA, eight of them per record), then sign-extended into the 64-bit accumulator --
  // 2^33 contributions per entry before it could overflow, quantisation 2^-26 of the level's largest gradient (the payload itself
  // carries 11 bits).  gmax = the level's largest |record value| as pass 1 saw it: the upstream gradients, and the totals of merged runs
  // (up to 64 lanes of a wavefront; round 6's first wave-wide merge kept the 16-lane bound of the row scans here and saturated
  // the conversion: tests/test_gpu_properties.py one_cell_const); + 1 % for the payload's fp16 rounding.
  const float fxs = fx_scale(gmax * 1.01f, 30);
  // one record = NW words, kept as ONE register tuple from its load to its use (separate scalars made the register allocator copy
  // the words out of the load's destination right behind the load, i.e. wait for it there)
  typedef uint32_t RecVec __attribute__((ext_vector_type(NW)));
  auto add = [&](const RecVec rv) {  // rv[0]: record key word (see PAIR RECORDS above)
    const uint32_t w0 = rv[0];
    // (the halfs are taken out of the payload WORDS here, with shifts: read through a half_t pointer the words were split into
    // 16-bit pieces where they are loaded)
    auto hv_at = [&](int j) -> half_t { return __builtin_bit_cast(half_t, (unsigned short)(rv[1 + (j >> 1)] >> (16 * (j & 1)))); };
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
      const float v = h2f(hv_at(j));
      if (v != 0.0f) {  // (many payloads ARE zero: w * g below the smallest fp16 -- dropping this test cost 8 % / 35 % at NV = 4 / 2)
        const float2_t p = float2_t{s0, s1} * v;  // one v_pk_mul_f32
        atomicAdd(a0 + j * seg, (unsigned long long)(long long)fx_round(p[0]));
        if (!single) atomicAdd(a1 + j * seg, (unsigned long long)(long long)fx_round(p[1]));
      }
    }
  };
  // this bin's eight lists inside the level's region (split records: key stream, then payload stream; RecWords)
  const uint32_t* lvl_keys = lists + lay.list_base[lvl] * NW;
  const uint32_t* lvl_pays = lvl_keys + (uint64_t)nbins * BS_LISTS * cap;
  const uint64_t bin_first = (uint64_t)b * BS_LISTS * cap;
  // fetch: ISSUES the loads of the window that starts at virtual position vw (a multiple of 64, wave-uniform: kept in scalar
  // registers, so that picking the window's list is SALU work) -- nothing here waits for them.  Lanes behind a list's last record read
  // the bin's first record and ignore it (an unconditional load: one under `if (ok)` is waited for at the end of that block -- the
  // merge with the default value is a copy).
  const uint32_t lane = threadIdx.x & 63u;
  auto fetch = [&](uint32_t vw, RecVec& rv, bool& ok) {
    uint32_t base_k = 0u, off_k = 0u, len_k = n0;
#define BS_PICK(k) if (vw >= o##k) { base_k = k##u * cap; off_k = o##k; len_k = n##k; }
    BS_PICK(1) BS_PICK(2) BS_PICK(3) BS_PICK(4) BS_PICK(5) BS_PICK(6) BS_PICK(7)
#undef BS_PICK
    const uint32_t r = vw - off_k + lane;
    ok = r < len_k;
    const uint64_t rec = bin_first + (ok ? base_k + r : 0u);
    if (RecWords<NV>::split) {  // one global_load_dword + one global_load_dwordx2
      uint32_t pay[NW - 1];
      const uint32_t key = lvl_keys[rec];
      __builtin_memcpy(pay, lvl_pays + rec * (NW - 1), (NW - 1) * sizeof(uint32_t));
      rv[0] = key;
#pragma unroll
      for (int q = 1; q < NW; ++q) rv[q] = pay[q - 1];
    } else {
      __builtin_memcpy(&rv, lvl_keys + rec * NW, NW * sizeof(uint32_t));  // one global_load_dwordx2
    }
  };
  // Two windows in flight: the records of window n + 1 are requested before window n is accumulated, two register sets in turn so that
  // no copy waits for a load.  The empty asm is where a window's records are waited for: BEFORE the next window is requested, so that
  // exactly one request is outstanding at every wait.
  RecVec rec_a = {}, rec_b = {};
  bool ok_a = false, ok_b = false;
  uint32_t vw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  const uint32_t vstep = blockDim.x;
  if (vw < v_end) {
    fetch(vw, rec_a, ok_a);
    for (;;) {
      asm volatile("" : "+v"(rec_a));
      vw += vstep;
      const bool more_b = vw < v_end;
      if (more_b) fetch(vw, rec_b, ok_b);
      if (ok_a) add(rec_a);
      if (!more_b) break;
      asm volatile("" : "+v"(rec_b));
      vw += vstep;
      const bool more_a = vw < v_end;
      if (more_a) fetch(vw, rec_a, ok_a);
      if (ok_b) add(rec_b);
      if (!more_a) break;
    }
  }
  if (n_ovf_mine != 0u) {  // records of this bin that did not fit their list (LISTS AND OVERFLOW): tagged with the bin in the level's overflow list
    const uint32_t n = min(ovf_cur[lvl], lay.ovf_cap[lvl]);
    const uint32_t* ov = ovf + lay.ovf_base[lvl] * (NW + 1);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      const uint32_t* rec = ov + (uint64_t)i * (NW + 1);
      if (rec[0] == (uint32_t)b) {
        RecVec rv;
#pragma unroll
        for (int q = 0; q < NW; ++q) rv[q] = rec[1 + q];
        add(rv);
      }
    }
  }
  __syncthreads();
  const double inv = (double)out_scale / (double)fxs;
  float* o = out + ((size_t)desc.offset[lvl] + lo) * NV;
  for (int i = threadIdx.x; i < n_el; i += blockDim.x) {  // i = entry * NV + j in the table's layout
    const long long vv = acc[(i % NV) * seg + i / NV];
    if (vv != 0) o[i] += (float)((double)vv * inv);  // sole owner of this segment
  }
}

// ---- host side ----------------------------------------------------------------------------------
// Entries per bin = 2^shift: the bin's int64 accumulators take 2^shift * NV * 8 bytes of LDS in pass 2 (64 KB -> two
// workgroups per CU), and more, smaller bins spread the pass-1 histogram atomics.  (NV = 2, the flow grid: 4,096-entry bins
// 1.01 ms in pass 2, 2,048: 0.89, 1,024: 0.93; NV = 4: 4,096-entry bins need one workgroup per CU: 1.84 -> 2.14 ms.)
static int bs_shift(int NV) { return NV == 4 ? 11 : NV == 2 ? 11 : 13; }

static inline bool bs_binned(const GridDesc& d, int l, int shift) {
  const int64_t nb = ((int64_t)d.size[l] + (1 << shift) - 1) >> shift;
  return ((d.hashed_mask >> l) & 1u) && is_pow2(d.size[l]) && nb <= BS_MAX_BINS && nb > 1;
}

BsPlan bs_plan(const GridDesc& d, int n_dims, int NV, int64_t P) {
  BsPlan pl;
  pl.shift = bs_shift(NV);
  pl.rec_words = 1 + (NV + 1) / 2;
  pl.n_wg = ceil_div64(P, BS_THREADS);
  // control block (zero-filled per call): level maxima, list cursors, overflow cursors, per-bin overflow counters
  pl.off_max = 0;
  pl.off_cur = 256;
  pl.off_ovf_cur = pl.off_cur + (int64_t)L4D_MAX_LEVELS * BS_LISTS * BS_MAX_BINS * 4;
  pl.off_ovf_cnt = pl.off_ovf_cur + L4D_MAX_LEVELS * 4;
  pl.ctrl_bytes = pl.off_ovf_cnt + (int64_t)L4D_MAX_LEVELS * BS_MAX_BINS * 4;
  pl.off_lists = (pl.ctrl_bytes + 255) / 256 * 256;
  uint64_t n_list = 0, n_ovf = 0;  // records
  for (int l = 0; l < L4D_MAX_LEVELS; ++l) {
    pl.lay.cap[l] = pl.lay.ovf_cap[l] = 0;
    pl.lay.list_base[l] = n_list;
    pl.lay.ovf_base[l] = n_ovf;
    if (l >= d.n_levels || !bs_binned(d, l, pl.shift)) continue;
    const int64_t nbins = ((int64_t)d.size[l] + (1 << pl.shift) - 1) >> pl.shift;
    // expected share of a list: pair records (2^(D-1) per sample) spread over bins x lists; + 25 % + 8 sigma + 64, a multiple of 32
    // records (list starts stay 128-byte aligned for 8- and 12-byte records)
    const double mean = (double)P * (double)(1 << (n_dims - 1)) / (double)(nbins * BS_LISTS);
    int64_t cap = (int64_t)(mean * 1.25 + 8.0 * sqrt(mean) + 64.0);
    cap = (cap + 31) / 32 * 32;
    const int64_t most = P << n_dims;  // a level cannot produce more records than this
    pl.lay.cap[l] = (uint32_t)std::min<int64_t>(cap, (most + 31) / 32 * 32);
    pl.lay.ovf_cap[l] = (uint32_t)std::min<int64_t>(most, 0xFFFFFFFFll);
    n_list += (uint64_t)nbins * BS_LISTS * pl.lay.cap[l];
    n_ovf += pl.lay.ovf_cap[l];
  }
  pl.off_ovf = (pl.off_lists + (int64_t)n_list * pl.rec_words * 4 + 255) / 256 * 256;
  pl.bytes = pl.off_ovf + (int64_t)n_ovf * (pl.rec_words + 1) * 4;
  return pl;
}

int bs_scatter(const GridDesc& desc, int n_dims, int NV, const float* x, int64_t P, int x_stride, const int* cols, const half_t* g,
               int g_stride, int g_col, float pre_scale, float* out, float out_scale, void* workspace, hipStream_t stream) {
  if (P == 0) return 0;
  const BsPlan pl = bs_plan(desc, n_dims, NV, P);
  char* ws = (char*)workspace;
  float* lvl_max = (float*)(ws + pl.off_max);
  uint32_t* cur = (uint32_t*)(ws + pl.off_cur);
  uint32_t* ovf_cur = (uint32_t*)(ws + pl.off_ovf_cur);
  uint32_t* ovf_cnt = (uint32_t*)(ws + pl.off_ovf_cnt);
  uint32_t* lists = (uint32_t*)(ws + pl.off_lists);
  uint32_t* ovf = (uint32_t*)(ws + pl.off_ovf);
  if ((P << n_dims) >= ((int64_t)1 << 32)) { l4d_set_error(1, "bs_scatter: too many points for one launch"); return 1; }
  l4d_fill_async(ws, 0u, pl.ctrl_bytes, stream);
  BsCols c;
  for (int d = 0; d < 3; ++d) c.c[d] = d < n_dims ? cols[d] : 0;
  for (int l = 0; l < desc.n_levels; ++l)  // the pair records carry their code in key bits 24..27: BINNED levels only (hashed,
    // power-of-two table, <= BS_MAX_BINS bins); larger tables never reach the binned path (atomic fallback) and need no limit
    if (bs_binned(desc, l, pl.shift) && desc.size[l] > (1u << 24)) { l4d_set_error(1, "bs_scatter: more than 2^24 entries in a binned level"); return 1; }
  int max_bins = 1;
  for (int l = 0; l < desc.n_levels; ++l) max_bins = std::max<int>(max_bins, (int)(((int64_t)desc.size[l] + (1 << pl.shift) - 1) >> pl.shift));
  max_bins = std::min(max_bins, BS_MAX_BINS);
  dim3 grid1((unsigned)xcd_grid(pl.n_wg));
  dim3 grid2(max_bins, desc.n_levels);
  const int lds2 = (1 << pl.shift) * NV * 8;
#define BS_LAUNCH(D, V)                                                                                                          \
  {                                                                                                                              \
    L4D_LAUNCH((bin_pass1_kernel<D, V>), grid1, dim3(BS_THREADS), 0, stream, desc, x, P, x_stride, c, g, g_stride, g_col, pre_scale, \
               pl.shift, (int64_t)pl.n_wg, pl.lay, cur, ovf_cur, ovf_cnt, lists, ovf, lvl_max, out, out_scale);                  \
    constexpr int DEF = V == 4 ? 11 : V == 2 ? 11 : 13; /* bs_shift(): compile-time bin size */                                  \
    (void)hipFuncSetAttribute((const void*)bin_reduce_kernel<D, V, DEF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);       \
    L4D_LAUNCH((bin_reduce_kernel<D, V, DEF>), grid2, dim3(1024), lds2, stream, desc, pl.shift, pl.lay, cur, ovf_cur, ovf_cnt, lists, \
               ovf, lvl_max, out, out_scale);                                                                                    \
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
