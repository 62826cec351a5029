/*
 * drs_dev.h -- development switches of libdrs_hip_dev.so (the same sources as libdrs_hip.so compiled with -DDRS_DEV).
 *
 * NOT part of the drop-in boundary (include/drs.h) and not exported by the product library.  They select between kernel forms
 * that are held bitwise equal by the tests (register-staged / LDS-DMA tiles, the cut of the filter gradient, skipping of the
 * all-halo taps) and feed the in-process A/B tools under tools/.  Every switch is PROCESS-GLOBAL state of the dev library:
 * all nets of a process that runs on the dev library share it.  Each setter returns the previous value; a negative argument
 * (where noted) only reads.
 */
#ifndef DRS_DEV_H_
#define DRS_DEV_H_
#ifdef __cplusplus
extern "C" {
#endif

int drs_debug_skip_taps(int v);          /* 0 multiply the all-halo taps / chunks too, 1 skip them where it pays (default), 2 always; < 0 reads */
int drs_debug_conv_variant(int v);       /* forward / input gradient: 0 register-staged tiles, 1 LDS-DMA halves, -1 per tile (default) */
int drs_debug_conv_wide192(int v);       /* Cout = 192 as one 128 x 192 tile (1, default) or three 128 x 64 tiles (0) */
int drs_debug_conv_lpt(int v);           /* plain forward / input-gradient launches: 1 start the full tiles first and the halo-skipping ones last (default), 0 natural order */
int drs_debug_conv_order(int B, int S, int k, int rate, int pad_before, int cin, int cout, int* out, int cap);   /* out[w] = tile of logical workgroup w of that plain launch; returns the workgroup count, 0 = natural order */
int drs_debug_conv_trace(void* dev_buffer); /* forward / input gradient (LDS-DMA form): device buffer [workgroups][2] of u64 that receives every workgroup's (start, end) on the 100 MHz real-time clock; NULL = off */
int drs_debug_conv_splitk(int v);        /* split-K of the forward / input-gradient pass: -1 by the cost model (default), 0 never, n >= 1 that many ranges */
int drs_debug_conv_hybrid(int v);        /* stream-K launches of more tiles than workgroups: 1 whole tiles for the full rounds, only the remainder cut (default), 0 every tile cut (r03) */
int drs_debug_conv_sk_order(int v);      /* a hybrid workgroup does 0 its range first, 1 its whole tiles first (default), 2 alternating by workgroup slot */
int drs_debug_conv_prio(int v);          /* forward / input gradient: waves lower their priority as their workgroup advances: -1 by the rule (stream-K launches from 20 K-steps per workgroup; default), 0 never, 1 always */
int drs_debug_conv_sk_geometry(int tiles, int nks, int bn, int* out3);   /* (workgroups, ranges, cut tiles) of a stream-K launch with the full workspace; returns workgroups, 0 = plain */
int drs_debug_wgrad_variant(int v);      /* filter gradient: 0 register-staged, 1 LDS-DMA halves, -1 per tile (default) */
int drs_debug_wgrad_seg(int v);          /* filter gradient, LDS-DMA form, sides >= 32 that are not a multiple of 32: 1 row-segment addressing without tables (default), 0 the table form */
int drs_debug_wgrad_balance(int v);      /* 1 cut the pixel dimension by live pixels (default), 0 equal chunk ranges */
int drs_debug_wgrad_target(int v);       /* workgroups the pixel split aims at (default 2048) */
int drs_debug_wgrad_target_big(int v);   /* the same on launches with many tiles and pixels (0 = default rule) */
int drs_debug_wgrad_len(int v);          /* chunks per workgroup the launches below the `big` class aim at (0 = default: 96, from 2^14 chunks 192) */
int drs_debug_wgrad_minchunks(int v);    /* fewest 32-pixel chunks a split of the pixel dimension may have (default 8) */
int drs_debug_wgrad_prio(int v);         /* filter gradient, wave priority by remaining work: -1 by the rule (default), 0 never, 1 levels 3..0, 2 levels 2..0 */
int drs_debug_wgrad_ablate(int v);       /* 1 = timing experiment (WRONG sums): every filter tap reads the un-shifted pixels (perfect X re-use); 2 = the S % 32 != 0 table reads of wgrad_dma_kernel right in front of each DMA issue, as before r04 (same sums); 3 = no wave priority by remaining work (same sums) */
int drs_debug_wgrad_model(int v);        /* 1 per-CU cost model for the workgroup count of launches below the `big` class (default), 0 the r02 table */
int drs_debug_slide_blocks(int v);       /* sliding elementwise kernels: workgroups the row-strip split aims at (default 5120) */
int drs_debug_slide_minrows(int v);      /* ... and the fewest rows of a strip in that first split (default 8) */
int drs_debug_jitter(unsigned long long seed);   /* schedule fuzzing of the two-stream training step: != 0 = sleeps of 0 .. 150 us (a generator seeded with it) on the step's streams where they hand work to each other; 0 = off */
int drs_debug_wg_stream_prio(int arm);   /* stream priority of the filter-gradient stream a net makes at its first two-stream step: 2 highest (default = the product), 0 the caller's level (as before round 5), 1 lowest and the collectives' side streams highest; < 0 reads */
int drs_debug_reductions_on_chain(int on);   /* 1: the classifier's slab reductions, the L2 term and the step's preparation launch stay on the compute stream (as before round 5); 0 (default = the product): on the filter-gradient stream */
int drs_debug_wgrad_schedule(int layer_slabs, int streams, int defer);   /* r06 experiment on the two-stream training step (no collectives): layer_slabs 1 (BEFORE drs_net_create: the net lists "gzL<i>" / "slabL<i>") = a gz slab and a split slab per layer, so the chain never waits for a filter gradient; streams 1..4 = layer i's filter gradient on stream i % streams; defer 0 = issued as the chain goes, 1 = all after the chain's last launch and behind it, 2 = issued after the chain's last launch, each behind its own gz only, 3 = issued as the chain goes but released by the end of its own block's input gradient (it then runs beside the NEXT block's elementwise passes); a negative argument leaves that setting */
int drs_debug_chain_mode(int v);         /* two-stream backward pass, the launches of the chain the step waits for at the top wave priority: -1 as the engine asks (default: 1 without collectives, 2 with), 0 none, 1 the input-gradient launches (+ stream-K fix-up), 2 + the batch-norm backward launches */
int drs_debug_slide_rowpad(int v);       /* TIMING EXPERIMENT (the callers must size z / idx / ga / gxh for it): phantom pixels after every stored image row in the two sliding elementwise kernels */
int drs_debug_cls_variant(int v);        /* classifier block: 1 MFMA from 4 classes up, LDS-DMA form up to C = 256 (default), 2 register MFMA form always, 3 LDS-DMA form where it fits, 0 vector-ALU always */
int drs_debug_variant(int v);            /* split-bf16 forward kernels: 0 register-staged, 1 LDS-DMA (default) */
/* the workgroups drs_conv_wgrad would launch for a shape, worked out on the host by the kernels' own assignment code:
   out[5 i ..] = (row tile, column tile, split, first chunk, end chunk) of workgroup i; out_tile_splits[r] = splits of row tile r.
   Returns the number of workgroups (nothing is written past cap), negative on a rejected shape. */
int drs_debug_wgrad_cut(int B, int S, int k, int rate, int pad_before, int cin, int cout, int* out, int cap, int* out_tile_splits,
                        int* out_tile_rows);

#ifdef __cplusplus
}
#endif
#endif /* DRS_DEV_H_ */
