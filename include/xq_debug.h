/*
 * xq_debug.h — diagnostic entry points of libxq_hip.so: build selectors, phase stamps and timing probes used by
 * tools/ (bench_tower.py, probe_tiles.py, probe_loop.py, compare_trunk_builds.py, soak_tower.py ...), by bench.py's
 * --tower-variant / --conv-variant / --search-occ switches and by the parity tests that compare two builds of one
 * kernel.  They replace nothing in the reference and are NOT part of the drop-in boundary (include/xq_selfplay.h):
 * no product code path calls them, every selector's default is the product build, and a caller that never touches
 * this header gets the same results.  Declared here so that every exported symbol of the library has a declaration
 * (tests/test_cabi_cpu.py asserts exports == xq_selfplay.h + xq_debug.h).
 */
#ifndef XQ_DEBUG_H
#define XQ_DEBUG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Build of the single-launch trunk kernel behind xq_tower_nhwc_bf16 (process-wide): -1 = automatic (the product:
 * k_tower1wa from 2,048 boards up, k_tower16b with 2 boards per workgroup below), 36 / 39 = k_tower16b with 2 / 4 boards per workgroup,
 * 60 = k_tower1wa (one wave per SIMD, hand-written layer body: the product from 2,048 boards up), 0 = k_tower (32x32x16
 * comparison build); every other value exists only in a -DXQ_TOWER_PROBES=1 library (experiments, ablations with wrong
 * results).  Builds 36, 39 and 60 compute the same bits. */
void xq_tower_set_variant(int variant);

/* Clock sample of the trunk kernel (process-wide; NULL = off, the default): while set, one workgroup in 64 of every
 * k_tower1wa launch through xq_tower_nhwc_bf16 adds its shader cycles (s_memtime), its 100 MHz ticks (s_memrealtime) and 1
 * to dev_u64x3[0..2] (device memory, zeroed by the caller).  cycles / ticks * 0.1 = the shader clock in GHz the chip held
 * while the kernel ran (MI355X_MICROARCH.md "DVFS give-back" item 6).  Results do not change. */
void xq_tower_set_clock_sample(void *dev_u64x3);

/* xq_tower_nhwc_bf16 with s_memtime phase stamps: 64 uint64 per workgroup into stamps_dev (0 start, 1 input conv
 * done, 2 its epilogue, 3+2L / 4+2L main loop / epilogue of layer L, 60 heads' MFMAs done, 61 end, 62 / 63
 * s_memrealtime at start / end).  No row map. */
int  xq_tower_debug_stamps(void *hip_stream, const void *planes_dev, const void *w1_dev, const void *wt_dev,
                           const void *bias_dev, const void *wh_dev, const void *bh_dev, void *policy_out_dev,
                           void *value_out_dev, int n_boards, int n_blocks, void *stamps_dev);

/* Bare MFMA / main-loop timing probes (results are meaningless; out_dev[1] = shader cycles, out_dev[2] =
 * s_memrealtime ticks of workgroup 0).  Returns XQ_E_INVALID in a library built without -DXQ_TOWER_PROBES=1. */
int  xq_mfma_probe(void *hip_stream, const void *seed64_dev, const void *weights_dev, void *out_dev, int n_workgroups,
                   int iters, int mode);

/* Build of the per-layer comparison kernel behind xq_conv3x3_nhwc_bf16: 1 = 2 boards per workgroup (default),
 * 2 = 4 boards per workgroup.  Same bits. */
void xq_conv3x3_set_variant(int variant);

/* xq_conv3x3_nhwc_bf16 (c_in = 128) with phase stamps, 32 uint64 per workgroup; ablate 1 = no stage barriers /
 * weight DMA, 2 = barriers only, 3 = DMA only (results wrong), 0 = the kernel as shipped. */
int  xq_conv3x3_debug_stamps(int variant, int ablate, void *hip_stream, const void *x_dev, const void *w_dev,
                             const void *bias_dev, const void *residual_dev, void *y_dev, int n_boards, int relu,
                             void *stamps_dev);

/* Kernel behind xq_policy_fc_bf16 (process-wide): -1 or 1 = k_policy_fc1w (one wave per SIMD, generated asm body: the product
 * since round 5), 0 = k_policy_fc (8 waves, HIP).  Same bits.  2..6 = other DMA placements and timing-only bodies of
 * k_policy_fc1w in a library built with -DXQ_TOWER_PROBES=1 (wrong results; XQ_E_INVALID from xq_policy_fc_bf16 otherwise). */
void xq_policy_fc_set_variant(int variant);

/* xq_policy_fc_bf16 with a timing-only body (wrong results): ablate 1 = no operand DMA behind the first two K-stages (what
 * the MFMA stream, its fragment reads and the stage barriers take), 2 = no MFMAs (what the L2 -> LDS operand delivery takes).
 * Returns XQ_E_INVALID in a library built without -DXQ_TOWER_PROBES=1. */
int  xq_policy_fc_debug(int ablate, void *hip_stream, const void *act_dev, const void *w_dev, const void *bias_dev, void *out_dev,
                        int n_rows, int n_cols, int k);

/* k_policy_fc1w with phase stamps: 8 uint64 per wave (32 per workgroup, workgroup = blockIdx) into stamps_dev: s_memtime at
 * start / first K-stage landed / K loop done / all stored, then s_memrealtime at start / end.  nodma != 0: the timing-only body
 * without operand DMA behind the prologue.  Returns XQ_E_INVALID in a library built without -DXQ_TOWER_PROBES=1. */
int  xq_policy_fc_debug_stamps(int nodma, void *hip_stream, const void *act_dev, const void *w_dev, const void *bias_dev,
                               void *out_dev, int n_rows, int n_cols, int k, void *stamps_dev);

/* Register budget of k_search_round as minimum waves per SIMD (process-wide): 4 (default, 128 VGPRs), or 3 / 5 / 6 /
 * 8; any other value means 4.  Same results. */
void xq_engine_set_search_occupancy(int waves_per_simd);

/* Phase stamps of k_search_round (probes library only; XQ_E_INVALID otherwise): a device buffer of 16 uint64 per game
 * that lane 0 of every game's wave fills with s_memtime at the phase boundaries of the round (0 start, 1 game record
 * loaded, 2 evaluator output consumed, 3 root board unpacked, 4 descent done, 5 leaf flags loaded, 6 path replayed,
 * 7 make_move done, 8 leaf record + planes stored, 9 dedupe insert done).  NULL switches them off.  Process-wide. */
int  xq_engine_set_search_stamps(void *dev_u64x16_per_game);

#ifdef __cplusplus
}
#endif
#endif
