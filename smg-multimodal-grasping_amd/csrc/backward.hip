// backward.hip - the backward walk: replaces loss.backward() of Trainer.backprop (code/trainer.py:350-351).
#include "engine.h"

#ifndef SMG_D3_TH8
#define SMG_D3_TH8 0      // dev A/B: 1 = the 3x3 data gradient of the big planes on 16 x 8 tiles, three workgroups per CU (round 6: built, parity green, measured SLOWER - DESIGN.md section 5.0)
#endif

// ------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------
// Pixel-chunk size of a weight-gradient launch: enough workgroups to fill the chip
// (~768) but no more - every workgroup ends with one fp32 atomicAdd per output element.
// Chunks are sized over the plane's VALID rows: every chunk starts inside [0, HW), so every workgroup of the launch stores its
// partial tile (a chunk that starts in the plane's padding rows - up to 127 of them since make_plane pads the big planes to 128
// rows - would leave without storing, and reduce_partials_kernel would add whatever the workspace held there).  The last chunk
// may run into the padding (the kernels clamp its length to HWp; padding rows hold zero gradients).
static void pick_chunk(const Plane& pl, int n_planes, int tiles_per_chunk, int& chunk, int& cps, int target = 768) {
    const int want = (target + tiles_per_chunk - 1) / tiles_per_chunk;
    cps = (want + n_planes - 1) / n_planes;
    if (cps < 1) cps = 1;
    chunk = ((pl.HW + cps - 1) / cps + 63) / 64 * 64;
    cps = (pl.HW + chunk - 1) / chunk;
}

// phases: bit 0 = the head and dense blocks 4, 3, 2 (down to the gradient of block 1's buffer), bit 1 = dense block 1, pool0 and
// the stem.  Between the two halves every gradient of [transition1 .. norm5] and of the head is final on `st`: a data-parallel
// caller starts their all-reduce there and hides it under the second half (smg_backward_phase).
int do_backward(smg_engine* e, const smg_net* net, const float* dq, hipStream_t st, int phases) {
    if (!e->have_fwd) return fail(-22, "smg_backward without a preceding smg_forward");
    if (!net->grads) return fail(-22, "net.grads is NULL");
    // phase bookkeeping: the second half continues the first half of the SAME forward (statistic arenas, ring position, G' buffers)
    if (phases == 2 && !e->bw_phase0_done)
        return fail(-22, "smg_backward_phase(1) must follow smg_backward_phase(0) of the same forward");
    if ((phases & 1) && e->bw_phase0_done)
        return fail(-22, "smg_backward / smg_backward_phase(0) while the second half of a two-phase backward is still due (run phase 1 or a new forward)");
    const Layout& L = *e->L;
    const TrunkRef& T = L.trunk[e->f_trunk];
    const HeadRef& Hd = L.head[e->f_head];
    const int NS = e->f_streams, NP = e->f_pairs;
    const float* P = net->params;
    float* Gr = net->grads;
    const Plane p4 = e->p_blk[3];
    const bool ph_a = phases & 1, ph_b = phases & 2;
    if (ph_a) {     // every replica (one 1-D fill each: the 2-D fill of the same bytes took 70 us per step)
        for (int r = 0; r < kStatRep; ++r) HIP_OK(hipMemsetAsync(e->bstat + (size_t)r * kStatRepStride, 0, 2 * e->bstat_span * sizeof(double), st));
    }
    const bool split16 = e->prec == 0 && kSplitOp == 3;      // the hot classes run on fp16-split operands: GS scaled by its recorded maximum,
                                                             // D2 written in unit form with per-block scales (bn_bwd_apply_split_kernel)
    if (ph_a && split16) HIP_OK(hipMemsetAsync(e->gamax, 0, (size_t)e->gamax_words * sizeof(unsigned), st));
    // Weight-gradient kernels only read what the data-gradient chain produces and write disjoint
    // gradient ranges, so they run on a second stream beside it (their MFMA/L2-bound phases overlap the
    // HBM-bound epilogues of the data-gradient kernels).  While profiling everything is serialised on
    // `st` so that per-kernel durations stay clean.
    // (a lowest-priority stream for the weight gradients gains 0.2 ms per step with one engine alive, and LOSES 10 ms as soon
    // as a second engine - two more streams - exists in the process: the streams then share hardware queues and serialise)
    const hipStream_t s2 = (e->prof || e->serialize) ? st : e->side;
    auto fork = [&](hipEvent_t ev) -> int {      // side stream continues after everything enqueued on st so far
        HIP_OK(hipEventRecord(ev, st));
        HIP_OK(hipStreamWaitEvent(s2, ev, 0));
        return 0;
    };
    // operand kind 3 needs the finished GS's recorded maximum (always materialised there); few streams: launch-bound, the fused form wins
    const bool gs_materialised = split16 || NS > 4;
    // ... and for calls of a few streams (the single-sample step of Trainer.backprop) the materialisation of the small planes (blocks 3-4,
    // precision mode 0) leaves the data-gradient CHAIN: the 3x3 data gradient applies the BN backward on load, scaled by its own halo's
    // maximum, and the side stream materialises GS (with the stream maximum its 3x3 weight gradient scales by) in front of the weight
    // gradients - one dependent 10-17 us launch less per layer on a chain that is all latency there (single-sample step 7.41-7.45 ->
    // 7.23-7.29 ms).  With 17 streams the side stream has no slack on those planes: the same move costs the headline step 0.15 ms
    // (16.44-16.50 -> 16.62-16.64 ms, alternating on one box), and on the big planes the two strided slice reads cost more than the launch.
    auto gs_on_side = [&](const Plane& pl) { return split16 && !e->generic3x3 && pl.HW <= 1600 && NS <= 4; };
    auto launch_gs_apply = [&](int b, int i, hipStream_t cs, float* GSb) {
        const Plane pl = e->p_blk[b];
        const int Ct = kBlockCtot[b];
        const DenseLayerRef& d = T.layers[b][i];
        BnBwdApplyArgs a{};
        a.g = e->G[b]; a.ldg = Ct; a.gcoff = d.cin; a.x = e->X[b]; a.ldx = Ct; a.xcoff = d.cin; a.pl = pl; a.C = kGrowth;
        a.xsum = fsum(e, e->st_X[b]); a.xsq = fsq(e, e->st_X[b]); a.xstride = Ct; a.xtab = stat_table(e, e->sx_tab[b], e->max_streams, Ct);
        a.s1 = b1(e, e->bs_X[b]); a.s2 = b2(e, e->bs_X[b]); a.sstride = Ct; a.scoff = d.cin; a.gamma = nullptr; a.eps = kEps;
        a.out = GSb; a.ldo = kGrowth; a.amax = split16 ? gamax_of(e, b, i, 0) : nullptr;
        BY(e, ESZ(e) * NS * pl.HW * 3 * kGrowth);
        ProfScope ps(e, cs, K_OTHER, 0);
        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_bwd_apply_kernel<PREC>), dim3((pl.HWp + bn_apply_rows(PREC) - 1) / bn_apply_rows(PREC), NS), dim3(256), 0, cs, a));
    };

    if (ph_a) {
    {   // value conv backward + relu1 + norm1 sums
        ValueBwdArgs a;
        a.h1 = e->H1; a.p4 = p4; a.hsum = fsum(e, e->st_H1); a.hsq = fsq(e, e->st_H1);
        a.gamma = P + Hd.n1.w; a.beta = P + Hd.n1.b; a.eps = kEps; a.w2p = e->packed_f + e->pk_head1;
        a.dq = dq; a.out_ch = e->head_out; a.OH = e->OH; a.OW = e->OW; a.dh1 = e->DH1;
        a.o1 = b1(e, e->bs_H1); a.o2 = b2(e, e->bs_H1); a.dbeta = Gr + Hd.n1.b; a.dgamma = Gr + Hd.n1.w; a.dw2 = Gr + Hd.c1.w;
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(value_bwd_kernel, dim3((p4.HW + 63) / 64, NP), dim3(256), 0, st, a);
    }
    int chunk4, cps4;
    pick_chunk(p4, NP, 2 * kFeat / 64, chunk4, cps4);
    {   // head conv0 weight gradient
        auto go = [&](auto ptag) -> int {
        BwdWeightP<CfgW64x64, W_ONE, C_IDENT, kPdWgrad, true, decltype(ptag)::value, true> p{};      // fp32 head buffers in every mode
        p.gbuf = e->DH1; p.ldg = kHeadMid; p.gcoff = 0; p.xbuf = e->H1; p.ldx = kHeadMid; p.xcoff = 0; p.pa = p4; p.MA = kHeadMid;
        p.xsum = fsum(e, e->st_H1); p.xsq = fsq(e, e->st_H1); p.xstride = kHeadMid;
        p.s1 = b1(e, e->bs_H1); p.s2 = b2(e, e->bs_H1); p.sstride = kHeadMid; p.scoff = 0; p.agamma = P + Hd.n1.w;
        p.bbuf = e->F; p.ldb = 2 * kFeat; p.pb = p4; p.NB = 2 * kFeat;
        p.bsum = fsum(e, e->st_F); p.bsq = fsq(e, e->st_F); p.bstride = 2 * kFeat; p.bgamma = P + Hd.n0.w; p.bbeta = P + Hd.n0.b;
        p.eps = kEps; p.chunk = chunk4; p.chunks_per_stream = cps4; p.n_chunks = NP * cps4;
        p.dw = Gr + Hd.c0.w; p.ldw_out = 2 * kFeat;
        if (fork(e->ev_misc)) return -5;
        BY(e, 4.0 * NP * p4.HW * (2 * kHeadMid + 2 * kFeat));
        return launch_wgrad(e, s2, p, dim3(1, 2 * kFeat / 64, NP * cps4), K_HW0, 2.0 * NP * p4.HW * 2 * kFeat * kHeadMid, 1, C_IDENT);
        };
        PREC_DISPATCH(e, if (int rc = go(PTAG)) return rc);
    }
    {   // head conv0 data gradient + relu0 + norm0 sums
        auto run = [&](auto tag, auto ptag) {
                using Cfg = decltype(tag);
                BwdDataP<Cfg, false, E_STORE, true, decltype(ptag)::value, true> p{};      // fp32 head buffers in every mode
        p.gbuf = e->DH1; p.ldg = kHeadMid; p.gcoff = 0; p.xbuf = e->H1; p.ldx = kHeadMid; p.xcoff = 0; p.pa = p4; p.KA = kHeadMid;
        p.xsum = fsum(e, e->st_H1); p.xsq = fsq(e, e->st_H1); p.xstride = kHeadMid;
        p.s1 = b1(e, e->bs_H1); p.s2 = b2(e, e->bs_H1); p.sstride = kHeadMid; p.scoff = 0; p.agamma = P + Hd.n1.w;
        p.wp = e->packed_u + e->pk_hd0; p.K8tot = kHeadMid / 8; p.ldn = 2 * kFeat; p.wcol0 = 0; p.N = 2 * kFeat;
        p.mbuf = e->F; p.ldm = 2 * kFeat; p.mcoff = 0; p.pm = p4;
        p.msum = fsum(e, e->st_F); p.msq = fsq(e, e->st_F); p.mstride = 2 * kFeat; p.egamma = P + Hd.n0.w; p.ebeta = P + Hd.n0.b;
        p.dst = e->DF; p.ldd = 2 * kFeat; p.dcoff = 0;
        p.o1 = b1(e, e->bs_F); p.o2 = b2(e, e->bs_F); p.ostride = 2 * kFeat; p.ocoff = 0;
        p.dbeta = Gr + Hd.n0.b; p.dgamma = Gr + Hd.n0.w; p.rep_stride = 0; p.eps = kEps;
        BY(e, 4.0 * NP * p4.HW * (2 * kHeadMid + 2 * 2 * kFeat));
        launch_gemm(e, st, p, dim3(NP * p4.HWp / Cfg::BM, 2 * kFeat / Cfg::BN), K_HD0, 2.0 * NP * p4.HW * 2 * kFeat * kHeadMid);
            };
            PREC_DISPATCH(e, if (p4.HWp % 128 == 0) run(CfgP128x128{}, PTAG); else run(CfgP64x128{}, PTAG));
    }
    {   // head norm0 backward + concat backward + norm5 backward -> G'_4
        Norm5BwdArgs a;
        a.DF = e->DF; a.F = e->F; a.p4 = p4; a.fsum = fsum(e, e->st_F); a.fsq = fsq(e, e->st_F);
        a.f1 = b1(e, e->bs_F); a.f2 = b2(e, e->bs_F); a.hgamma = P + Hd.n0.w;
        a.x4 = e->X[3]; a.xsum = fsum(e, e->st_X[3]); a.xsq = fsq(e, e->st_X[3]); a.gamma5 = P + T.norm5.w; a.eps = kEps;
        a.user_ptr = e->d_user_ptr; a.user_pair = e->d_user_pair; a.user_slot = e->d_user_slot;
        a.G4 = e->G[3]; a.SA = b1(e, e->bs_X[3]); a.SB = b2(e, e->bs_X[3]);
        a.dbeta5 = Gr + T.norm5.b; a.dgamma5 = Gr + T.norm5.w; a.chunk = 16;
        ProfScope ps(e, st, K_OTHER, 0);
        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(norm5_bwd_kernel<PREC>), dim3(4, NS, (p4.HW + 15) / 16), dim3(256), 0, st, a));
    }
    }   // ph_a: head
    // Buffers of a dense layer's finished gradients (GS: its 32 output channels, D2: its bottleneck, D2S: D2's block scales): a ring of
    // kRing slots in backward order, which the side stream releases as its weight gradients finish.  The ring position of a layer is
    // static (a two-phase backward's second half continues where the first stopped).
    // (Measured and rejected in round 5: own buffers for the layers of blocks 3-4 and their weight gradients DEFERRED to the
    // bandwidth-bound kernels of block 2 - the idea being that a concurrent kernel stretches the 10-20 us chain kernels of the small
    // planes.  Same box, alternating: 16.44-16.66 ms per step without, 17.13-17.40 with four deferred layers per layer of block 2,
    // 17.37-17.50 with eight: the side stream's work on the small planes is what fills an otherwise idle chip.)
    struct LayerBuf { float* GS; float* D2; float* D2S; int ring; };
    auto buf_of = [&](int b, int i) -> LayerBuf {
        const int r = ring_pos(b, i);
        return LayerBuf{e->GS[r % kRing], e->D2[r % kRing], e->D2S[r % kRing], r};
    };
    // The two weight gradients of dense layer (b, i) and their shared reduce, on the side stream (which must already wait for the
    // layer's D2 / GS).
    // Few-stream calls without "deterministic" (the 1x1 weight gradient adds with atomics, only the 3x3 one leaves partial tiles): the
    // reduce of layer l waits for layer l - 1's and ONE launch serves both - the partial tiles alternate between the two halves of the
    // workspace.  (The side stream is the longer one in a single-sample step: 29 reduce launches of ~7.5 us less on it.)
    ReduceArgs pend3{}; pend3.Z = 0;
    int pend3_half = 0;
    auto flush_pending_reduce = [&]() { if (pend3.Z) { launch_reduce2(e, s2, K_W3, pend3, ReduceArgs{}); pend3.Z = 0; } };
    auto issue_wgrads = [&](int b, int i) -> int {
        const Plane pl = e->p_blk[b];
        const int Ct = kBlockCtot[b];
        const DenseLayerRef& d = T.layers[b][i];
        float* bt = el(e, e->Bt, e->bt_off[b][i]);
        const LayerBuf lb = buf_of(b, i);
        GradSrc gsrc{};
        if (gs_on_side(pl)) launch_gs_apply(b, i, s2, lb.GS);
        if (gs_materialised) { gsrc.g = lb.GS; gsrc.ldg = kGrowth; gsrc.amax = gamax_of(e, b, i, 0); }
        else {
            gsrc.g = el(e, e->G[b], d.cin); gsrc.ldg = Ct; gsrc.x = el(e, e->X[b], d.cin); gsrc.ldx = Ct;
            gsrc.xsum = fsum(e, e->st_X[b]) + d.cin; gsrc.xsq = fsq(e, e->st_X[b]) + d.cin;
            gsrc.s1 = b1(e, e->bs_X[b]) + d.cin; gsrc.s2 = b2(e, e->bs_X[b]) + d.cin; gsrc.sstride = Ct; gsrc.eps = kEps;
        }
        ReduceArgs red3{}; red3.Z = 0;            // the 3x3 weight gradient's reduction, launched together with the 1x1 one below
        int64_t part3_floats = 0;                 // ... and the partial-tile floats it occupies
        if (!e->generic3x3) {
            // conv2 weight gradient with the activation halo resident in LDS (halo.cuh)
            const int ts = halo_tile(pl, NS);
            Halo3x3WgradArgs a;
            a.g = gsrc; a.pl = pl; a.src = bt; a.C = kBottleneck;
            const int th = 8;                              // tiles are ts x 8 pixels
            a.bt = bn_table(e, e->sb_tab[b][i], e->max_streams, 0, kBottleneck, P + d.n2.w, P + d.n2.b);
            a.asc = asc_n2(e, b, i);
            a.tiles_x = (pl.W + ts - 1) / ts; a.n_tiles = ((pl.H + th - 1) / th) * a.tiles_x;
            a.tiles_per_wg = w3_tiles_per_wg(a.n_tiles, ts, NS, e->part_floats, (double)ts / th);
            const int groups = (a.n_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
            if ((int64_t)groups * NS * 9 * 32 * kBottleneck > e->part_floats) return fail(-12, "partial-gradient workspace too small");
            const bool pair_reduce = !e->deterministic && NS <= 4 && (int64_t)groups * NS * 9 * 32 * kBottleneck * 2 <= e->part_floats;
            if (!pair_reduce) flush_pending_reduce();
            float* part3 = e->part + (pair_reduce && pend3_half ? e->part_floats / 2 : 0);
            a.part = part3;
            a.groups = groups; a.streams = NS;
            const unsigned w3_grid = (unsigned)(((groups * NS + 7) / 8) * 8 * (kBottleneck / 32));      // (see the kernel: channel groups of a tile group share an XCD)
            {
                BY(e, ESZ(e) * NS * pl.HW * (kGrowth + kBottleneck));
                ProfScope ps(e, s2, K_W3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth);
                if (ts == 16) {
                    PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_wgrad_kernel<16, PREC>), dim3(w3_grid), dim3(256),
                                                        (HaloWgradSGeo<16, PREC>::smem_bytes()), s2, a));
                } else {
                    PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_wgrad_kernel<8, PREC>), dim3(w3_grid), dim3(256),
                                                        (HaloWgradSGeo<8, PREC>::smem_bytes()), s2, a));
                }
            }
            red3.part = part3; red3.Z = groups * NS; red3.taps = 9; red3.rows = kGrowth; red3.cols = kBottleneck; red3.ldp = kBottleneck;
            red3.z_stride = (int64_t)9 * kGrowth * kBottleneck; red3.tap_stride = (int64_t)kGrowth * kBottleneck;
            red3.dw = Gr + d.c2.w; red3.ldw_out = kBottleneck * 9; red3.cmap = C_3x3;
            part3_floats = (int64_t)groups * NS * 9 * kGrowth * kBottleneck;
            if (pair_reduce) {       // this layer's reduce rides with the next layer's (or the final flush)
                if (pend3.Z) { launch_reduce2(e, s2, K_W3, pend3, red3); pend3.Z = 0; }
                else pend3 = red3;
                pend3_half ^= 1;
                red3.Z = 0; part3_floats = 0;
            }
        } else {   // conv2 weight gradient (generic implicit GEMM, one launch slice per tap)
            flush_pending_reduce();
            const int chunk = 512, cps = (pl.HWp + chunk - 1) / chunk;   // latency-bound: many short workgroups
            BwdWeightP<CfgW32x128, W_THREE, C_3x3, kPdWgrad, false> p{};
            p.gbuf = lb.GS; p.ldg = kGrowth; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.MA = kGrowth;
            p.bbuf = bt; p.ldb = kBottleneck; p.pb = pl; p.NB = kBottleneck;
            p.bsum = fsum(e, e->st_Bt[b][i]); p.bsq = fsq(e, e->st_Bt[b][i]); p.bstride = kBottleneck;
            p.bgamma = P + d.n2.w; p.bbeta = P + d.n2.b; p.eps = kEps;
            p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
            p.dw = Gr + d.c2.w; p.ldw_out = kBottleneck * 9;
            BY(e, ESZ(e) * NS * pl.HW * (kGrowth + kBottleneck));
            if (int rc = launch_wgrad(e, s2, p, dim3(1, 1, 9 * NS * cps), K_W3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth, 9, C_3x3)) return rc;
        }
        // conv1 weight gradient, wave-specialised (wsw.cuh): 128 x 128 tiles, 32-pixel k-tiles, loader + matrix waves - for the partial-tile
        // form (more than four streams, or "deterministic") on layers of more than 64 input channels (a half-empty 128-column tile
        // costs block 1's first layer 77 -> 107 us; few-stream calls keep the generic kernel's 128 x 64 atomics: 7.3 -> 7.5 ms per single-sample step)
        if (split16 && !e->generic_w1 && d.cin > 64 && (e->deterministic || NS > 4)) {
            using G = WswGeo;
            const int nt = (d.cin + G::BN - 1) / G::BN;
            int chunk, cps;
            pick_chunk(pl, NS, nt, chunk, cps, 320);      // (256 / 512 / 768 workgroups: 16.45-16.62 / 16.64-16.67 / 16.73-16.82 ms per step against 16.47-16.53)
            Wgrad1x1WsArgs a{};
            a.d2 = reinterpret_cast<const u32x4*>(lb.D2); a.binv = lb.D2S; a.x = e->X[b]; a.ldb = Ct; a.pl = pl; a.NB = d.cin;
            a.btab = stat_table(e, e->sx_tab[b], e->max_streams, Ct); a.bgamma = P + d.n1.w; a.bbeta = P + d.n1.b; a.basc = asc_n1(e, b, i);
            a.chunk = chunk; a.chunks_per_stream = cps; a.n_chunks = NS * cps;
            a.dw = Gr + d.c1.w; a.ldw_out = d.cin; a.ldp = nt * G::BN;
            const bool w1_part = e->deterministic || NS > 4;
            const int64_t need = (int64_t)a.n_chunks * G::BM * a.ldp;
            int64_t off1 = part3_floats;
            if (red3.Z && (!w1_part || off1 + need > e->part_floats)) { launch_reduce2(e, s2, K_W3, red3, ReduceArgs{}); red3.Z = 0; off1 = 0; }
            a.part = (w1_part && off1 + need <= e->part_floats) ? e->part + off1 : nullptr;
            if (w1_part && !a.part && e->deterministic)
                return fail(-12, "deterministic: a weight-gradient launch needs " + std::to_string(need) + " partial-tile floats, the workspace holds " + std::to_string(e->part_floats - off1));
            a.tm = TileMap{a.n_chunks, nt, 0};
            const size_t smem = G::smem_bytes(chunk);
            static bool raised[64] = {};
            if (!raised[e->device & 63]) {
                (void)hipFuncSetAttribute((const void*)conv1x1_wgrad_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                raised[e->device & 63] = true;
            }
            {
                BY(e, 4.0 * NS * pl.HW * (kBottleneck + d.cin));
                ProfScope ps(e, s2, K_W1, 2.0 * NS * pl.HW * d.cin * kBottleneck);
                hipLaunchKernelGGL(conv1x1_wgrad_ws_kernel, dim3(tile_grid(a.tm)), dim3(512), smem, s2, a);
            }
            ReduceArgs red1{}; red1.Z = 0;
            if (a.part) {
                red1.part = a.part; red1.Z = a.n_chunks; red1.taps = 1; red1.rows = G::BM; red1.cols = d.cin; red1.ldp = a.ldp;
                red1.z_stride = (int64_t)G::BM * a.ldp; red1.tap_stride = (int64_t)a.n_chunks * G::BM * a.ldp;
                red1.dw = a.dw; red1.ldw_out = d.cin; red1.cmap = C_IDENT;
            }
            launch_reduce2(e, s2, K_W1, red3, red1);
        } else {   // conv1 weight gradient.  ~320 workgroups: it shares the chip with the data-gradient chain on the other stream
            // (256..384 measure the same, 512 / 768 / 1024 cost the step 0.15 / 0.35 / 0.75 ms)
            using Cfg = CfgW128x64;
            const int nt = (d.cin + Cfg::BN - 1) / Cfg::BN;
            int chunk, cps;
            // 16-bit storage: the k-loop is a third as long, the 128 x 64 atomics per workgroup are not - half as many workgroups
            // on many-stream batches (config 3: 28.9 -> 28.5 ms at 160; 120 / 80: 28.6 / 28.8; S = 1824 with 5 streams: 320 stays)
            // few-stream calls on the atomics form: 128 (every workgroup adds a 128 x 64 tile with fp32 atomics; single-sample step 5.9 -> 5.65 ms;
            // 64 / 192 / 320: 6.0 / 5.7 / 5.9)
            pick_chunk(pl, NS, nt, chunk, cps, (e->prec && NS >= 16) ? 160 : (NS <= 4 && !e->deterministic) ? 128 : 320);
            auto go = [&](auto ptag) -> int {
                BwdWeightP<MC<Cfg, decltype(ptag)::value>, W_ONE, C_IDENT, kPdWgrad, false, decltype(ptag)::value> p{};
                p.gbuf = lb.D2; p.ldg = kBottleneck; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.MA = kBottleneck; p.binv = lb.D2S; p.basc = asc_n1(e, b, i);
                p.bbuf = e->X[b]; p.ldb = Ct; p.pb = pl; p.NB = d.cin;
                p.bsum = fsum(e, e->st_X[b]); p.bsq = fsq(e, e->st_X[b]); p.bstride = Ct; p.btab = stat_table(e, e->sx_tab[b], e->max_streams, Ct);
                p.bgamma = P + d.n1.w; p.bbeta = P + d.n1.b; p.eps = kEps;
                p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
                p.dw = Gr + d.c1.w; p.ldw_out = d.cin;
                BY(e, ESZ(e) * NS * pl.HW * (kBottleneck + d.cin));
                // partial tiles + the fixed-order reduce (reproducible; since reduce_partials splits the partials over four waves it
                // beats 128 x 64 fp32 atomics per workgroup); a few streams: host-launch-bound, atomics save the reduce launches
                const bool w1_part = e->deterministic || NS > 4;
                // both reductions of the layer in ONE launch: the 1x1 partial tiles go behind the 3x3 ones (if they fit; else the 3x3
                // reduction runs first and the workspace is reused)
                ReduceArgs red1{}; red1.Z = 0;
                int64_t off1 = part3_floats;
                if (red3.Z && (!w1_part || off1 + (int64_t)NS * cps * Cfg::BM * nt * Cfg::BN > e->part_floats)) { launch_reduce2(e, s2, K_W3, red3, ReduceArgs{}); red3.Z = 0; off1 = 0; }
                if (int rc = launch_wgrad(e, s2, p, dim3(1, nt, NS * cps), K_W1, 2.0 * NS * pl.HW * d.cin * kBottleneck, 1, C_IDENT, w1_part, off1, &red1)) return rc;
                launch_reduce2(e, s2, K_W1, red3, red1);
                return 0;
            };
            PREC_DISPATCH(e, if (int rc = go(PTAG)) return rc);
        }
        return 0;
    };
    // debug_stop (engine.h): leave the backward here - both streams joined, the forward's state consumed
    auto debug_stop = [&]() -> int {
        flush_pending_reduce();
        HIP_OK(hipEventRecord(e->ev_end, s2));
        HIP_OK(hipStreamWaitEvent(st, e->ev_end, 0));
        HIP_OK(hipGetLastError());
        e->have_fwd = false; e->bw_phase0_done = false; e->prof_stage = -1;
        return 0;
    };
    for (int b = ph_a ? 3 : 0; b >= (ph_b ? 0 : 1); --b) {
        e->prof_stage = b;
        const Plane pl = e->p_blk[b];
        const int Ct = kBlockCtot[b];
        if (e->dbg_stop == b * 100 + 50) return debug_stop();
        for (int i = (int)T.layers[b].size() - 1; i >= 0; --i) {
            const DenseLayerRef& d = T.layers[b][i];
            float* bt = el(e, e->Bt, e->bt_off[b][i]);
            const LayerBuf lb = buf_of(b, i);
            float* GSb = lb.GS;
            float* D2b = lb.D2;
            // ring slots: at every third layer wait for the side stream's event of the LATEST of the next three slots' previous users
            // (ring position + 2 - kRing; the side stream is in order, so the two before it are done as well)
            static_assert(kRing % 3 == 0 && kRing > 3, "the sparse ring wait covers three slots at a time");
            if (lb.ring >= kRing && lb.ring % 3 == 0) HIP_OK(hipStreamWaitEvent(st, e->ev_side[(lb.ring + 2) % kRing], 0));
            // This layer's finished output-slice gradient GS = invstd*(G' - SA/n - xhat*SB/n), materialised once (dense
            // [px][32]) for the 3x3 data- and weight-gradient kernels.  They can also apply it while loading the G' / X
            // slices (GradSrc with x set): one launch less on the dependency chain, but measured 0.5 ms per step slower on
            // many-stream batches - two strided 128-B-per-pixel reads replace one dense one in both consumers.
            GradSrc gsrc{};
            gsrc.g = el(e, e->G[b], d.cin); gsrc.ldg = Ct; gsrc.x = el(e, e->X[b], d.cin); gsrc.ldx = Ct;
            gsrc.xsum = fsum(e, e->st_X[b]) + d.cin; gsrc.xsq = fsq(e, e->st_X[b]) + d.cin;
            gsrc.s1 = b1(e, e->bs_X[b]) + d.cin; gsrc.s2 = b2(e, e->bs_X[b]) + d.cin; gsrc.sstride = Ct; gsrc.eps = kEps;
            if (e->generic3x3 || (gs_materialised && !gs_on_side(pl))) {
                launch_gs_apply(b, i, st, GSb);
                if (gs_materialised) { gsrc = GradSrc{}; gsrc.g = GSb; gsrc.ldg = kGrowth; gsrc.amax = gamax_of(e, b, i, 0); }
            }
            if (!e->generic3x3) {
                // conv2 (3x3) data gradient with the gradient halo resident in LDS (halo.cuh)
                Halo3x3DgradArgs a;
                a.g = gsrc; a.pl = pl; a.C = kBottleneck;
                a.mbuf = bt;
                a.dst = split16 ? e->DY2 : D2b; a.o1 = b1(e, e->bs_Bt[b][i]); a.o2 = b2(e, e->bs_Bt[b][i]); a.ostride = kBottleneck;
                a.wu = e->packed_u + e->pk_hd[b][i]; a.bt = bn_table(e, e->sb_tab[b][i], e->max_streams, 0, kBottleneck, P + d.n2.w, P + d.n2.b);
                BY(e, ESZ(e) * NS * pl.HW * (kGrowth + 2 * kBottleneck));      // gradient in, mask source in, dy out
                ProfScope ps(e, st, K_D3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth);
                const bool th8 = SMG_D3_TH8 && halo_tile(pl, NS) == 16 && e->prec == 0 && kSplitOp == 3 && pl.H % 8 == 0 && pl.W % 16 == 0;
                TraceScope ts(st, K_D3, th8 ? dim3((pl.H / 8) * ((pl.W + 15) / 16), NS)
                                            : halo_tile(pl, NS) == 16 ? dim3(((pl.H + 15) / 16) * ((pl.W + 15) / 16), NS) : dim3(((pl.H + 7) / 8) * ((pl.W + 7) / 8), NS, kBottleneck / 64));
                if (halo_tile(pl, NS) == 16) {
                    static bool raised[64][3] = {};          // the 16x16 kernel needs more than the default 64 KB of dynamic LDS
                    if (!raised[e->device & 63][e->prec]) {
                        PREC_DISPATCH(e, (void)hipFuncSetAttribute((const void*)conv3x3_halo_dgrad_kernel<16, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, (HaloDgradSGeo<16, PREC>::smem_bytes(kBottleneck))));
                        PREC_DISPATCH(e, (void)hipFuncSetAttribute((const void*)conv3x3_halo_dgrad_kernel<16, PREC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (HaloDgradSGeo<16, PREC>::smem_bytes(kBottleneck))));
                        raised[e->device & 63][e->prec] = true;
                    }
                    a.tiles_x = (pl.W + 15) / 16; a.cg_per_wg = kBottleneck / 32;
                    if (th8) {
                        // 16 x 8 tiles (halo.cuh): one MFMA tile per wave, 52.7 KB of LDS - three workgroups per CU instead of two
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_dgrad_kernel<16, 0, false, 8>), dim3((pl.H / 8) * a.tiles_x, NS), dim3(256),
                                           (HaloDgradSGeo<16, 0, 8>::smem_bytes(kBottleneck)), st, a);
                    } else
                    if (pl.H % 16 || pl.W % 16) {      // tiles hang over the edge: the bounds-checked instantiation
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_dgrad_kernel<16, PREC, true>), dim3(((pl.H + 15) / 16) * a.tiles_x, NS), dim3(256),
                                           (HaloDgradSGeo<16, PREC>::smem_bytes(kBottleneck)), st, a));
                    } else {
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_dgrad_kernel<16, PREC>), dim3((pl.H / 16) * a.tiles_x, NS), dim3(256),
                                           (HaloDgradSGeo<16, PREC>::smem_bytes(kBottleneck)), st, a));
                    }
                } else {
                    static bool raised8[64][3] = {};         // two buffers of a whole kernel row's weights: past the default 64 KB in the fp32-class mode
                    if (!raised8[e->device & 63][e->prec]) {
                        PREC_DISPATCH(e, (void)hipFuncSetAttribute((const void*)conv3x3_halo_dgrad_kernel<8, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, (HaloDgradSGeo<8, PREC>::smem_bytes(kBottleneck))));
                        raised8[e->device & 63][e->prec] = true;
                    }
                    a.tiles_x = (pl.W + 7) / 8; a.cg_per_wg = 1;      // small planes: one 64-channel group per workgroup
                    PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_dgrad_kernel<8, PREC>), dim3(((pl.H + 7) / 8) * a.tiles_x, NS, kBottleneck / 64), dim3(256),
                                       (HaloDgradSGeo<8, PREC>::smem_bytes(kBottleneck)), st, a));
                }
            } else {   // conv2 (3x3) data gradient -> dy of relu2/norm2 (D2) + norm2 sums (generic implicit GEMM)
                auto run = [&](auto tag) {
                    using Cfg = decltype(tag);
                    BwdDataP<Cfg, true, E_STORE, false> p{};
                    p.gbuf = GSb; p.ldg = kGrowth; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.KA = kGrowth;
                    p.wp = e->packed_u + e->pk_g3d[b][i]; p.K8tot = 9 * kGrowth / 8; p.ldn = kBottleneck; p.wcol0 = 0; p.N = kBottleneck;
                    p.mbuf = bt; p.ldm = kBottleneck; p.mcoff = 0; p.pm = pl;
                    p.msum = fsum(e, e->st_Bt[b][i]); p.msq = fsq(e, e->st_Bt[b][i]); p.mstride = kBottleneck;
                    p.egamma = P + d.n2.w; p.ebeta = P + d.n2.b;
                    p.dst = split16 ? e->DY2 : D2b; p.ldd = kBottleneck; p.dcoff = 0;
                    p.o1 = b1(e, e->bs_Bt[b][i]); p.o2 = b2(e, e->bs_Bt[b][i]); p.ostride = kBottleneck; p.ocoff = 0;
                    p.dbeta = Gr + d.n2.b; p.dgamma = Gr + d.n2.w; p.eps = kEps;
                    BY(e, ESZ(e) * NS * pl.HW * (kGrowth + 2 * kBottleneck));
                    launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, kBottleneck / Cfg::BN), K_D3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth);
                };
                if (pl.HWp % 128 == 0) run(CfgP128x128{}); else run(CfgP64x128{});
            }
            if (split16) {   // norm2 backward applied once: D2 (unit form, per-block scales) <- gamma2*invstd*(dy - s1/n - xhat*s2/n)
                BnBwdApplySplitArgs a{};
                a.g = e->DY2; a.x = bt; a.pl = pl;
                a.xsum = fsum(e, e->st_Bt[b][i]); a.xsq = fsq(e, e->st_Bt[b][i]); a.xstride = kBottleneck; a.xtab = !e->generic3x3 ? stat_table(e, e->sb_tab[b][i], e->max_streams, kBottleneck) : StatTab{};
                a.s1 = b1(e, e->bs_Bt[b][i]); a.s2 = b2(e, e->bs_Bt[b][i]); a.sstride = kBottleneck;
                a.gamma = P + d.n2.w; a.eps = kEps; a.out = reinterpret_cast<u32x4*>(D2b); a.binv = lb.D2S;
                if (!e->generic3x3) { a.dbeta = Gr + d.n2.b; a.dgamma = Gr + d.n2.w; }   // the halo dgrad leaves these to us
                BY(e, 4.0 * NS * pl.HW * 3 * kBottleneck);
                ProfScope ps(e, st, K_OTHER, 0);
                hipLaunchKernelGGL(bn_bwd_apply_split_kernel, dim3(pl.HWp / kScaleBlock, NS), dim3(256), 0, st, a);
            } else {   // 16-bit modes: norm2 backward applied once, in place: D2 <- gamma2*invstd*(dy - s1/n - xhat*s2/n)
                BnBwdApplyArgs a{};
                a.g = D2b; a.ldg = kBottleneck; a.gcoff = 0; a.x = bt; a.ldx = kBottleneck; a.xcoff = 0; a.pl = pl; a.C = kBottleneck;
                a.xsum = fsum(e, e->st_Bt[b][i]); a.xsq = fsq(e, e->st_Bt[b][i]); a.xstride = kBottleneck; a.xtab = !e->generic3x3 ? stat_table(e, e->sb_tab[b][i], e->max_streams, kBottleneck) : StatTab{};
                a.s1 = b1(e, e->bs_Bt[b][i]); a.s2 = b2(e, e->bs_Bt[b][i]); a.sstride = kBottleneck; a.scoff = 0;
                a.gamma = P + d.n2.w; a.eps = kEps; a.out = D2b; a.ldo = kBottleneck; a.amax = nullptr;
                if (!e->generic3x3) { a.dbeta = Gr + d.n2.b; a.dgamma = Gr + d.n2.w; }   // the halo dgrad leaves these to us
                BY(e, ESZ(e) * NS * pl.HW * 3 * kBottleneck);
                ProfScope ps(e, st, K_OTHER, 0);
                PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_bwd_apply_kernel<PREC>), dim3((pl.HWp + bn_apply_rows(PREC) - 1) / bn_apply_rows(PREC), NS), dim3(256), 0, st, a));
            }
            // the layer's weight gradients: ONE fork per layer, behind the norm2 apply (GS and D2 are both final there)
            if (fork(e->ev_d2[lb.ring % kRing])) return -5;
            if (int rc = issue_wgrads(b, i)) return rc;
            HIP_OK(hipEventRecord(e->ev_side[lb.ring % kRing], s2));
            // conv1 (1x1) data gradient -> relu1/norm1 backward accumulated into G'.  Layers are grouped (kGroup,
            // from the top of the block): inside a group only the channels the group itself produced - needed by
            // the very next layer - are accumulated per layer; everything below the group's lowest layer is done
            // once for the whole group by BwdDataGroupP (gemm.cuh), which touches G' and x once instead of once
            // per layer.
            if (e->dbg_stop == b * 100 + i && e->prec == 0) {      // tests: G' of the block in front of this layer's 1x1 data gradients
                const int64_t nf = (int64_t)NS * pl.HWp * Ct;
                if (e->dbg_gsnap_floats < nf) {
                    if (e->dbg_gsnap) (void)hipFree(e->dbg_gsnap);
                    e->dbg_gsnap = nullptr; e->dbg_gsnap_floats = 0;
                    HIP_OK(hipMalloc((void**)&e->dbg_gsnap, (size_t)nf * sizeof(float)));
                    e->dbg_gsnap_floats = nf;
                }
                HIP_OK(hipMemcpyAsync(e->dbg_gsnap, e->G[b], (size_t)nf * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            const int L = (int)T.layers[b].size();
            const int g_lo = i - ((L - 1 - i) % kGroup == kGroup - 1 ? 0 : std::min(i, kGroup - 1 - (L - 1 - i) % kGroup));
            const int cs = T.layers[b][g_lo].cin;                       // channels below the group
            if (d.cin > cs) {                                           // [cs, cin): per-layer accumulate
                auto run = [&](auto tag, auto ptag) {
                    using Cfg = MC<decltype(tag), decltype(ptag)::value>;
                    BwdDataP<Cfg, false, E_ACCUM, false, decltype(ptag)::value> p{};
                    p.gbuf = D2b; p.ldg = kBottleneck; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.KA = kBottleneck; p.binv = lb.D2S;
                    p.wp = e->packed_u + e->pk_d1[b][i]; p.K8tot = kBottleneck / 8; p.ldn = d.cin; p.wcol0 = cs; p.N = d.cin - cs;
                    p.mbuf = e->X[b]; p.ldm = Ct; p.mcoff = cs; p.pm = pl;
                    p.msum = fsum(e, e->st_X[b]); p.msq = fsq(e, e->st_X[b]); p.mstride = Ct; p.mtab = stat_table(e, e->sx_tab[b], e->max_streams, Ct);
                    p.egamma = P + d.n1.w + cs; p.ebeta = P + d.n1.b + cs;
                    p.dst = e->G[b]; p.ldd = Ct; p.dcoff = cs;
                    p.o1 = b1(e, e->bs_X[b]); p.o2 = b2(e, e->bs_X[b]); p.ostride = Ct; p.ocoff = cs;
                    p.dbeta = e->dbscr + e->db_off[b][i] + cs; p.dgamma = e->dbscr + e->db_off[b][i] + d.cin + cs; p.rep_stride = e->db_total; p.eps = kEps;
                    BY(e, ESZ(e) * NS * pl.HW * (kBottleneck + 3.0 * p.N));          // dy in; x in, G' read + written
                    launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, (p.N + Cfg::BN - 1) / Cfg::BN), K_D1, 2.0 * NS * pl.HW * p.N * kBottleneck);
                };
                PREC_DISPATCH(e, if (pl.HWp % 128 == 0) run(CfgP128x64{}, PTAG); else run(CfgP64x64{}, PTAG));
            }
            if (i == g_lo) {                                            // [0, cs): the whole group at once
                auto run = [&](auto tag, auto ptag) {
                    using Cfg = MC<decltype(tag), decltype(ptag)::value>;
                    BwdDataGroupP<Cfg, decltype(ptag)::value> p{};
                    const int g_hi = L - 1 - ((L - 1 - g_lo) / kGroup) * kGroup;      // top layer of this group
                    p.nseg = g_hi - g_lo + 1;
                    for (int k = 0; k < p.nseg; ++k) {
                        const DenseLayerRef& dk = T.layers[b][g_lo + k];
                        const LayerBuf lk = buf_of(b, g_lo + k);
                        p.seg[k].g = lk.D2; p.seg[k].wp = e->packed_u + e->pk_d1[b][g_lo + k]; p.seg[k].ldn = dk.cin; p.seg[k].binv = lk.D2S;
                        p.seg[k].gamma = P + dk.n1.w; p.seg[k].beta = P + dk.n1.b;
                        p.seg[k].dbeta = e->dbscr + e->db_off[b][g_lo + k]; p.seg[k].dgamma = e->dbscr + e->db_off[b][g_lo + k] + dk.cin;
                    }
                    p.ldg = kBottleneck; p.pa = pl; p.KA = kBottleneck; p.N = cs;
                    p.mbuf = e->X[b]; p.ldm = Ct;
                    p.msum = fsum(e, e->st_X[b]); p.msq = fsq(e, e->st_X[b]); p.mstride = Ct; p.mtab = stat_table(e, e->sx_tab[b], e->max_streams, Ct);
                    p.dst = e->G[b]; p.ldd = Ct;
                    p.o1 = b1(e, e->bs_X[b]); p.o2 = b2(e, e->bs_X[b]); p.ostride = Ct; p.rep_stride = e->db_total; p.eps = kEps;
                    BY(e, ESZ(e) * NS * pl.HW * ((double)p.nseg * kBottleneck + 3.0 * cs));
                    launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, (cs + Cfg::BN - 1) / Cfg::BN), K_D1, 2.0 * NS * pl.HW * cs * kBottleneck * p.nseg);
                };
                // mode 0, 128-row tiles: 32-deep k-tiles with ONE tile of loads in flight - the staging registers of two 16-deep
                // tiles, half the barriers (k-loop 65k -> 50k cycles per workgroup, launches -7 %; same k16 order, same bits)
                // (MC doubles the depth in the 16-bit modes: 128 x 64 x 64 / 64 x 64 x 64)
                PREC_DISPATCH(e, if (pl.HWp % 128 == 0) run(GemmCfg<128, 64, 32, 2, 2, 1, true>{}, PTAG); else run(CfgP64x64{}, PTAG));
            }
            if (e->dbg_stop == b * 100 + i) return debug_stop();
        }
        if (b > 0) {   // transition b-1: X[b-1] (all channels) -> X[b][:, 0:C0]
            const Plane pp = e->p_blk[b - 1];
            const int Cp = kBlockCtot[b - 1], C0 = kBlockCin[b];
            {
                int chunk, cps;
                pick_chunk(pl, NS, (C0 / 128) * (Cp / 128), chunk, cps);
                auto go = [&](auto ptag) -> int {
                BwdWeightP<MC<CfgW128x128, decltype(ptag)::value>, W_POOL, C_IDENT, 1, true, decltype(ptag)::value> p{};      // (one k-tile in flight: the pooling fetch holds 4 float4 per slot, three tiles of them leave one workgroup per CU)
                p.gbuf = e->G[b]; p.ldg = Ct; p.gcoff = 0; p.xbuf = e->X[b]; p.ldx = Ct; p.xcoff = 0; p.pa = pl; p.MA = C0;
                p.xsum = fsum(e, e->st_X[b]); p.xsq = fsq(e, e->st_X[b]); p.xstride = Ct;
                p.s1 = b1(e, e->bs_X[b]); p.s2 = b2(e, e->bs_X[b]); p.sstride = Ct; p.scoff = 0; p.agamma = nullptr;
                p.bbuf = e->X[b - 1]; p.ldb = Cp; p.pb = pp; p.NB = Cp;
                p.bsum = fsum(e, e->st_X[b - 1]); p.bsq = fsq(e, e->st_X[b - 1]); p.bstride = Cp;
                p.bgamma = P + T.tnorm[b - 1].w; p.bbeta = P + T.tnorm[b - 1].b; p.eps = kEps;
                p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
                p.dw = Gr + T.tconv[b - 1].w; p.ldw_out = Cp;
                if (fork(e->ev_misc)) return -5;
                BY(e, ESZ(e) * NS * (2.0 * pl.HW * C0 + (double)pp.HW * Cp));
                flush_pending_reduce();      // (the partial-tile workspace is about to be reused from its start)
                return launch_wgrad(e, s2, p, dim3(C0 / 128, Cp / 128, NS * cps), K_TW, 2.0 * NS * pl.HW * Cp * C0, 1, C_IDENT);
                };
                PREC_DISPATCH(e, if (int rc = go(PTAG)) return rc);
            }
            if (pp.H != 2 * pl.H || pp.W != 2 * pl.W) {
                ProfScope ps(e, st, K_OTHER, 0);
                PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(zero_uncovered_kernel<PREC>), dim3(256, NS), dim3(256), 0, st, (void*)e->G[b - 1], Cp, pp, 2 * pl.H, 2 * pl.W, Cp));
            }
            {
                auto run = [&](auto tag, auto ptag) {
                using Cfg = MC<decltype(tag), decltype(ptag)::value>;
                BwdDataP<Cfg, false, E_UNPOOL, true, decltype(ptag)::value> p{};
                p.gbuf = e->G[b]; p.ldg = Ct; p.gcoff = 0; p.xbuf = e->X[b]; p.ldx = Ct; p.xcoff = 0; p.pa = pl; p.KA = C0;
                p.xsum = fsum(e, e->st_X[b]); p.xsq = fsq(e, e->st_X[b]); p.xstride = Ct;
                p.s1 = b1(e, e->bs_X[b]); p.s2 = b2(e, e->bs_X[b]); p.sstride = Ct; p.scoff = 0; p.agamma = nullptr;
                p.wp = e->packed_u + e->pk_td[b - 1]; p.K8tot = C0 / 8; p.ldn = Cp; p.wcol0 = 0; p.N = Cp;
                p.mbuf = e->X[b - 1]; p.ldm = Cp; p.mcoff = 0; p.pm = pp;
                p.msum = fsum(e, e->st_X[b - 1]); p.msq = fsq(e, e->st_X[b - 1]); p.mstride = Cp;
                p.egamma = P + T.tnorm[b - 1].w; p.ebeta = P + T.tnorm[b - 1].b;
                p.dst = e->G[b - 1]; p.ldd = Cp; p.dcoff = 0;
                p.o1 = b1(e, e->bs_X[b - 1]); p.o2 = b2(e, e->bs_X[b - 1]); p.ostride = Cp; p.ocoff = 0;
                p.dbeta = e->dbscr + e->db_toff[b - 1]; p.dgamma = e->dbscr + e->db_toff[b - 1] + Cp; p.rep_stride = e->db_total; p.eps = kEps;
                BY(e, ESZ(e) * NS * (2.0 * pl.HW * C0 + 2.0 * pp.HW * Cp));
                launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, Cp / Cfg::BN), K_TD, 2.0 * NS * pl.HW * Cp * C0);
            };
            PREC_DISPATCH(e, run(CfgP64x128{}, PTAG));   // the 128-row variant of the unpool epilogue spills registers
            }
        }
    }
    e->prof_stage = -1;
    if (ph_b) {
    {   // pool0 / relu0 backward + norm0 sums
        Pool0BwdArgs a;
        a.G1 = e->G[0]; a.X1 = e->X[0]; a.ld1 = kBlockCtot[0]; a.p1 = e->p_blk[0];
        a.xsum = fsum(e, e->st_X[0]); a.xsq = fsq(e, e->st_X[0]); a.xstride = kBlockCtot[0];
        a.SA = b1(e, e->bs_X[0]); a.SB = b2(e, e->bs_X[0]); a.sstride = kBlockCtot[0];
        a.argmax = e->argmax; a.stem = e->stem; a.ps = e->p_stem;
        a.ssum = fsum(e, e->st_stem); a.ssq = fsq(e, e->st_stem);
        a.gamma = P + T.norm0.w; a.beta = P + T.norm0.b; a.eps = kEps;
        a.DY0 = e->DY0; a.o1 = b1(e, e->bs_stem); a.o2 = b2(e, e->bs_stem);
        a.dbeta = Gr + T.norm0.b; a.dgamma = Gr + T.norm0.w;
        ProfScope ps(e, st, K_OTHER, 0);
        if (e->p_stem.H % 8 || e->p_stem.W % 8) return fail(-22, "stem plane must tile by 8 (input_size multiple of 16)");
        a.tiles_per_wg = 8;          // 4..20 measure the same; 1 costs 0.7 ms per step in atomics
        const int n_t = (e->p_stem.H / 8) * (e->p_stem.W / 8);
        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(pool0_bwd_kernel<PREC>), dim3((n_t + a.tiles_per_wg - 1) / a.tiles_per_wg, NS), dim3(256), 0, st, a));
    }
    {   // conv0 weight gradient (no data gradient: the image needs none)
        const Plane ps_ = e->p_stem;
        int chunk, cps;
        pick_chunk(ps_, NS, 1, chunk, cps);
        auto go = [&](auto ptag, auto mtag) -> int {
        constexpr int SM = decltype(mtag)::value;      // W_STEM (3-channel image, 196 columns) or W_STEM1 (one channel, 49 taps, replicated x3 at the flush)
        using WCfg = typename std::conditional<SM == W_STEM1, CfgW64x64, CfgW64x256>::type;
        BwdWeightP<WCfg, SM, SM == W_STEM1 ? C_STEM1 : C_STEM, kPdWgrad, true, decltype(ptag)::value> p{};      // (fp32 image / stem plane in every mode)
        p.gbuf = e->DY0; p.ldg = 64; p.gcoff = 0; p.xbuf = e->stem; p.ldx = 64; p.xcoff = 0; p.pa = ps_; p.MA = 64;
        p.xsum = fsum(e, e->st_stem); p.xsq = fsq(e, e->st_stem); p.xstride = 64;
        p.s1 = b1(e, e->bs_stem); p.s2 = b2(e, e->bs_stem); p.sstride = 64; p.scoff = 0; p.agamma = P + T.norm0.w;
        p.bbuf = e->img4; p.ldb = 4; p.pb = e->p_img; p.NB = SM == W_STEM1 ? 64 : 196;
        p.eps = kEps; p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
        p.dw = Gr + T.conv0.w; p.ldw_out = 147;
        if (fork(e->ev_misc)) return -5;
        BY(e, 4.0 * NS * (2.0 * ps_.HW * 64 + (double)e->p_img.HW * (SM == W_STEM1 ? 1 : 4)));
        flush_pending_reduce();      // (the partial-tile workspace is about to be reused from its start)
        return launch_wgrad(e, s2, p, dim3(1, 1, NS * cps), K_SW, 2.0 * NS * ps_.HW * 64 * 147, 1, SM == W_STEM1 ? C_STEM1 : C_STEM);
        };
        if (e->f_stem1) { PREC_DISPATCH(e, if (int rc = go(PTAG, std::integral_constant<int, W_STEM1>{})) return rc); }
        else { PREC_DISPATCH(e, if (int rc = go(PTAG, std::integral_constant<int, W_STEM>{})) return rc); }
    }
    }   // ph_b: pool0 + stem
    {   // the trunk's dbeta / dgamma: replicas -> gradient array (and zeroed for the next call).  Each half flushes ONLY the segments
        // it produced - the table holds dense block 1's 2 x 6 segments first, then blocks 2-4 and the transitions - because the
        // flush is a non-atomic read-modify-write of the gradient array: between the halves of a two-phase backward the ranges
        // [trunk split, end) + head are being all-reduced IN PLACE on another stream (parallel.OverlappedGradSync), and the second
        // half must not touch a single element of them (it would re-write a stale local value over the reduced one).
        const int first_b = 2 * kBlockLayers[0];
        const int seg0 = ph_b ? 0 : first_b, seg1 = ph_a ? e->n_dbseg : first_b;
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(db_flush_kernel, dim3(4, seg1 - seg0 + 1), dim3(256), 0, st,
                           reinterpret_cast<const DbSegD*>(e->d_dbseg + (size_t)e->f_trunk * e->n_dbseg) + seg0,
                           e->dbscr, e->db_total, kDbRep, Gr,
                           ph_b ? b1(e, e->bs_stem) : nullptr, ph_b ? b2(e, e->bs_stem) : nullptr, NS, Gr + T.norm0.b, Gr + T.norm0.w);
    }
    flush_pending_reduce();
    HIP_OK(hipEventRecord(e->ev_end, s2));          // join: everything after the backward (or this half of it) sees every gradient
    HIP_OK(hipStreamWaitEvent(st, e->ev_end, 0));
    HIP_OK(hipGetLastError());
    e->bw_phase0_done = phases == 1;
    return 0;
}

