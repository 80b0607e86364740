// forward.hip - the forward walk over DenseNet-121: replaces reinforcement_net.forward / reactive_net.forward
// (code/models.py:361-586, :72-296) for one (trunk, head).
#include "engine.h"

#ifndef SMG_C1WS_BN
#define SMG_C1WS_BN 128     // dev A/B: 64 = two 64-column workgroups per 64-row tile (rounds 3-5)
#endif
#ifndef SMG_FWD16_WS
#define SMG_FWD16_WS 0
#endif

// Host-side checks of a batch against the engine (index ranges, capacities).
int validate_batch(const smg_engine* e, const smg_batch* B) {
    const int NS = B->n_streams, NP = B->n_pairs;
    if (NS < 1 || NS > e->max_streams || NP < 1 || NP > e->max_pairs) return fail(-22, "batch exceeds engine capacity");
    if (!B->images_nchw_dev && !B->heightmaps_dev) return fail(-22, "no input images");
    for (int s = 0; s < NS; ++s)
        if (B->stream_image[s] < 0 || B->stream_image[s] >= B->n_images) return fail(-22, "stream_image out of range");
    if (B->masks_dev) {
        if (!B->heightmaps_dev || !B->stream_mask_a || !B->stream_mask_b) return fail(-22, "device masks need the heightmap input form and both index arrays");
        for (int s = 0; s < NS; ++s)
            if (B->stream_mask_a[s] >= B->n_masks || B->stream_mask_b[s] >= B->n_masks || (B->stream_mask_a[s] < 0 && B->stream_mask_b[s] >= 0))
                return fail(-22, "stream mask index out of range");
    }
    for (int j = 0; j < NP; ++j)
        if (B->pair_a[j] < 0 || B->pair_a[j] >= NS || B->pair_b[j] < 0 || B->pair_b[j] >= NS) return fail(-22, "pair index out of range");
    const int n_seq_t = B->bn_seq_trunk ? B->n_bn_seq_trunk : 0, n_seq_h = B->bn_seq_head ? B->n_bn_seq_head : 0;
    const int R = e->max_streams > e->max_pairs ? e->max_streams : e->max_pairs;
    if (n_seq_t > 4 * R + 16 || n_seq_h > 4 * R + 16) return fail(-22, "bn sequence too long");
    if (B->heightmaps_dev) {
        const int pad = (e->S - 2 * B->hm_size) / 2;
        if (pad < 0 || 2 * B->hm_size + 2 * pad != e->S) return fail(-22, "heightmap size does not match engine input_size");
    }
    return 0;
}

// The batch description as the device block holds it (engine.h so_*), written to a pinned host block `h`.
void fill_stage(const smg_engine* e, const smg_batch* B, int* h) {
    const int NS = B->n_streams, NP = B->n_pairs;
    const int n_seq_t = B->bn_seq_trunk ? B->n_bn_seq_trunk : 0, n_seq_h = B->bn_seq_head ? B->n_bn_seq_head : 0;
    memcpy(h + e->so_image, B->stream_image, NS * sizeof(int));
    memcpy(h + e->so_rot, B->stream_rotated, NS * sizeof(int));
    memcpy(h + e->so_aff, B->stream_affine, 6 * NS * sizeof(float));
    if (B->masks_dev) {
        memcpy(h + e->so_ma, B->stream_mask_a, NS * sizeof(int));
        memcpy(h + e->so_mb, B->stream_mask_b, NS * sizeof(int));
    }
    memcpy(h + e->so_pa, B->pair_a, NP * sizeof(int));
    memcpy(h + e->so_pb, B->pair_b, NP * sizeof(int));
    if (n_seq_t) memcpy(h + e->so_seq_t, B->bn_seq_trunk, n_seq_t * sizeof(int));
    if (n_seq_h) memcpy(h + e->so_seq_h, B->bn_seq_head, n_seq_h * sizeof(int));
    // users of each stream's features (CSR), for the backward
    int* ptr = h + e->so_uptr; int* up = h + e->so_upair; int* us = h + e->so_uslot; int n = 0;
    ptr[0] = 0;
    for (int s = 0; s < NS; ++s) {
        for (int j = 0; j < NP; ++j) {
            if (B->pair_a[j] == s) { up[n] = j; us[n] = 0; ++n; }
            if (B->pair_b[j] == s) { up[n] = j; us[n] = 1; ++n; }
        }
        ptr[s + 1] = n;
    }
}

int do_forward(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* B,
                      float* q_out, hipStream_t st) {
    const Layout& L = *e->L;
    const TrunkRef& T = L.trunk[trunk_id];
    const HeadRef& Hd = L.head[head_id];
    const int NS = B->n_streams, NP = B->n_pairs;
    if (int rc = validate_batch(e, B)) return rc;
    const int n_seq_t = B->bn_seq_trunk ? B->n_bn_seq_trunk : 0, n_seq_h = B->bn_seq_head ? B->n_bn_seq_head : 0;
    const int pad = B->heightmaps_dev ? (e->S - 2 * B->hm_size) / 2 : 0;

    if (e->capturing) {     // graph capture: the caller filled h_stage_g (fill_stage); the copy becomes a node that re-reads it at every replay
        HIP_OK(hipMemcpyAsync(e->d_stage, e->h_stage_g, (size_t)e->stage_ints * sizeof(int), hipMemcpyHostToDevice, st));
    } else {   // batch description -> device: no host synchronisation on the forward path
        const int turn = e->stage_turn; e->stage_turn ^= 1;
        HIP_OK(hipEventSynchronize(e->ev_stage[turn]));   // the copy issued two forwards ago (long done)
        fill_stage(e, B, e->h_stage[turn]);
        HIP_OK(hipMemcpyAsync(e->d_stage, e->h_stage[turn], (size_t)e->stage_ints * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_OK(hipEventRecord(e->ev_stage[turn], st));
    }
    if (kFStatRep > 1) HIP_OK(hipMemset2DAsync(e->fstat, kStatRepStride * sizeof(double), 0, 2 * e->fstat_span * sizeof(double), kFStatRep, st));
    else HIP_OK(hipMemsetAsync(e->fstat, 0, 2 * e->fstat_span * sizeof(double), st));

    // weights -> K-major packs
    {
        const unsigned n_pack = (unsigned)(e->h_pack[trunk_id].size() + e->h_pack_head[head_id].size());
        if (e->prec == 0 && kSplitOp == 3) {       // scales of the fp16-split operands: weight headers + activation scales (gemm.cuh, operand kind 3)
            ProfScope ps(e, st, K_OTHER, 0);
            hipLaunchKernelGGL(scale_kernel, dim3(1, n_pack + (unsigned)e->n_asc), dim3(1024), 0, st,
                               e->d_pack + (trunk_id * 3 + head_id) * e->pack_stride, (int)n_pack, e->d_asc + (trunk_id * 3 + head_id) * e->n_asc,
                               net->params, e->packed_u, e->asc, e->prec);
        }
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(pack_weights_kernel, dim3(64, n_pack), dim3(256), 0, st,
                           e->d_pack + (trunk_id * 3 + head_id) * e->pack_stride, net->params, e->packed_u, e->packed_f, e->prec);
    }
    const float* P = net->params;

    // The trunk of streams [s0, s0 + ns).  Streams are independent up to the head (BN statistics are per stream), so
    // the batch is run as TWO chains on two HIP streams: the tail of one chain's kernel overlaps the other chain's
    // next kernel (two full sweeps side by side take 83 % of their serial time, tests/gpu_concurrency_probe.py).
    // BN table of one consumer layer (rows [r0, r0 + rows) of a [max rows][C] table at float offset `at`) + the launch that fills it
    // table entries of channels [c0, c0 + C) whose producer is not a dense layer (block inputs, head features)
    auto bn_stat = [&](hipStream_t cs, const BnTab& t, int rows, const double* sum, const double* sq, int sstride, int c0, int C, int count) {
        BnStatArgs a;
        a.sum = sum; a.sq = sq; a.sstride = sstride; a.eps = kEps; a.inv_count = 1.0 / (double)count;
        a.mean = const_cast<float*>(t.mean); a.invstd = const_cast<float*>(t.invstd); a.ld = t.ld; a.c0 = c0; a.C = C; a.rows = rows;
        ProfScope ps(e, cs, K_OTHER, 0);
        hipLaunchKernelGGL(bn_stat_kernel, dim3((rows * C + 255) / 256), dim3(256), 0, cs, a);
    };
    // Units [u_lo, u_hi) of the chain: unit 0 = input preparation + stem + pool0, then one unit per dense layer and per
    // transition.  The caller alternates the chains unit by unit, so that both have work queued from the start (a chain
    // enqueued whole keeps the host busy for ~1.3 ms, during which the other chain's HIP stream sits empty).
    auto trunk_chain = [&](const int s0, const int ns, hipStream_t cs, const int u_lo, const int u_hi) -> int {
        int unit = 0;
        auto on = [&]() { const bool r = unit >= u_lo && unit < u_hi; ++unit; return r; };
        auto xs = [&](int b) { return el(e, e->X[b], (int64_t)s0 * e->p_blk[b].HWp * kBlockCtot[b]); };
        auto st_off = [&](double* base, int stride) { return base + (int64_t)s0 * stride; };
        const bool stem1 = B->heightmaps_dev != nullptr;                      // heightmap form: the three channels are identical by construction
        float* img4 = stem1 ? e->img4 + (int64_t)s0 * e->p_img.HWp : e->img4 + (int64_t)s0 * e->p_img.HWp * 4;
        float* stem = e->stem + (int64_t)s0 * e->p_stem.HWp * 64;
        const bool head_unit = on();
        if (head_unit) {   // K1 input preparation
            PrepArgs a;
            a.images_nchw = B->images_nchw_dev; a.heightmaps = B->heightmaps_dev; a.hm = B->hm_size; a.pad = pad; a.S = e->S;
            a.mean = B->image_mean; a.stdv = B->image_std;
            a.stream_image = e->d_stream_image + s0; a.stream_affine = e->d_affine + 6 * s0; a.stream_rotated = e->d_stream_rot + s0;
            a.img4 = img4; a.img1 = stem1 ? img4 : nullptr; a.HWp = e->p_img.HWp;
            a.masks = B->masks_dev; a.stream_mask_a = e->d_stage + e->so_ma + s0; a.stream_mask_b = e->d_stage + e->so_mb + s0;
            ProfScope ps(e, cs, K_OTHER, 0);
            hipLaunchKernelGGL(prep_rotate_kernel, dim3((e->S * e->S + 255) / 256, ns), dim3(256), 0, cs, a);
        }
        if (head_unit) {   // stem conv0 7x7/2
            auto run = [&](auto tag, auto ptag, auto mtag) {
                using Cfg = decltype(tag);
                constexpr int SM = decltype(mtag)::value;          // F_STEM (3-channel image) or F_STEM1 (one channel, 49 taps)
                FwdConvP<Cfg, SM, decltype(ptag)::value> p{};
                p.src = img4; p.lds_ = 4; p.ps = e->p_img; p.po = e->p_stem; p.K = 0;
                p.wp = e->packed_u + (SM == F_STEM1 ? e->pk_conv0_1 : e->pk_conv0); p.K8tot = (SM == F_STEM1 ? 64 : 224) / 8; p.N = 64;
                p.dst = stem; p.ldd = 64; p.dcoff = 0;
                p.dsum = st_off(fsum(e, e->st_stem), 64); p.dsq = st_off(fsq(e, e->st_stem), 64); p.dstride = 64;
                BY(e, 4.0 * ns * ((double)e->p_img.HW * (SM == F_STEM1 ? 1 : 4) + (double)e->p_stem.HW * 64));
                launch_gemm(e, cs, p, dim3(ns * e->p_stem.HWp / Cfg::BM, 1), K_STEM, 2.0 * ns * e->p_stem.HW * 64 * 147);
            };
            if (stem1) { PREC_DISPATCH(e, if (e->p_stem.HWp % 128 == 0) run(CfgP128x64{}, PTAG, std::integral_constant<int, F_STEM1>{}); else run(CfgP64x64{}, PTAG, std::integral_constant<int, F_STEM1>{})); }
            else { PREC_DISPATCH(e, if (e->p_stem.HWp % 128 == 0) run(CfgP128x64{}, PTAG, std::integral_constant<int, F_STEM>{}); else run(CfgP64x64{}, PTAG, std::integral_constant<int, F_STEM>{})); }
        }
        if (head_unit) {   // norm0 + relu0 + pool0
            Pool0Args a;
            a.stem = stem; a.ps = e->p_stem; a.ssum = st_off(fsum(e, e->st_stem), 64); a.ssq = st_off(fsq(e, e->st_stem), 64);
            a.gamma = P + T.norm0.w; a.beta = P + T.norm0.b; a.eps = kEps;
            a.x1 = xs(0); a.ldx = kBlockCtot[0]; a.po = e->p_blk[0];
            a.dsum = st_off(fsum(e, e->st_X[0]), kBlockCtot[0]); a.dsq = st_off(fsq(e, e->st_X[0]), kBlockCtot[0]); a.dstride = kBlockCtot[0];
            a.argmax = e->argmax + (int64_t)s0 * e->p_blk[0].HWp * 64;
            ProfScope ps(e, cs, K_OTHER, 0);
            PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(pool0_kernel<PREC>), dim3(e->p_blk[0].HWp / 64, ns), dim3(256), 0, cs, a));
        }
        for (int b = 0; b < 4; ++b) {
            e->prof_stage = b;
            const Plane pl = e->p_blk[b];
            const int Ct = kBlockCtot[b];
            double* xsum = st_off(fsum(e, e->st_X[b]), Ct); double* xsq = st_off(fsq(e, e->st_X[b]), Ct);
            for (size_t i = 0; i < T.layers[b].size(); ++i) {
                if (!on()) continue;
                const DenseLayerRef& d = T.layers[b][i];
                float* bt = el(e, e->Bt, e->bt_off[b][i] + (int64_t)s0 * pl.HWp * kBottleneck);
                double* bsum = st_off(fsum(e, e->st_Bt[b][i]), kBottleneck); double* bsq = st_off(fsq(e, e->st_Bt[b][i]), kBottleneck);
                {   // norm1 + relu + conv1 (1x1, cin -> 128)
                    const BnTab t1 = bn_table(e, e->sx_tab[b], e->max_streams, s0, Ct, P + d.n1.w, P + d.n1.b);
                    if (i == 0) bn_stat(cs, t1, ns, xsum, xsq, Ct, 0, d.cin, pl.HW);     // block input: from pool0 / the transition
                    auto run_p = [&](auto tag, auto ptag) {
                        using Cfg0 = decltype(tag);
                        using Cfg = MC<Cfg0, decltype(ptag)::value, (Cfg0::BK < 32 ? 2 : 1)>;      // (cin is a multiple of 32 only)
                        FwdConvP<Cfg, F_ONE, decltype(ptag)::value> p{};
                        p.src = xs(b); p.lds_ = Ct; p.ps = pl; p.po = pl; p.K = d.cin; p.asc = asc_n1(e, b, (int)i);
                        p.bt = t1; p.fresh0 = i == 0 ? d.cin : d.cin - kGrowth; p.fsum = xsum; p.fsq = xsq; p.fstride = Ct; p.eps = kEps;
                        p.tw_mean = const_cast<float*>(t1.mean); p.tw_invstd = const_cast<float*>(t1.invstd);
                        p.wp = e->packed_u + e->pk_c1[b][i]; p.K8tot = d.cin / 8; p.N = kBottleneck;
                        p.dst = bt; p.ldd = kBottleneck; p.dcoff = 0;
                        p.dsum = bsum; p.dsq = bsq; p.dstride = kBottleneck;
                        BY(e, ESZ(e) * ns * pl.HW * (d.cin + kBottleneck));
                        launch_gemm(e, cs, p, dim3(ns * pl.HWp / Cfg::BM, kBottleneck / Cfg::BN), K_C1, 2.0 * ns * pl.HW * d.cin * kBottleneck);
                    };
                    auto run = [&](auto tag) { PREC_DISPATCH(e, run_p(tag, PTAG)); };
                    // 128x128 tiles where the plane tiles by 128 rows and the launch still fills the chip; else 64x64 (BK = 32) -
                    // and when even that leaves most CUs idle (few streams per call, or the 20x20 planes), 32x64 tiles with
                    // the k-tile split over wave pairs: twice the workgroups, half the serial K chain.
                    const int small_wgs = 320;      // (160 / 640: 24.4 / 24.6 ms per step against 24.6; 512 / 1024 slower on the 17-stream step)
                    const int wg128 = ns * pl.HWp / 128, wg64 = ns * pl.HWp / 64 * 2;
                    if (!e->generic_c1 && e->prec == 0 && !(pl.HWp % 128 == 0 && wg128 >= small_wgs) && wg64 >= small_wgs && d.cin % 32 == 0 && pl.HWp % 64 == 0) {
                        // wave-specialised 64 x 64 x 32 (ws.cuh)
                        Fwd1x1WsArgs a{};
                        a.src = xs(b); a.lds_ = Ct; a.pl = pl; a.K = d.cin;
                        a.bt = t1; a.fresh0 = i == 0 ? d.cin : d.cin - kGrowth; a.fsum = xsum; a.fsq = xsq; a.fstride = Ct; a.eps = kEps;
                        a.tw_mean = const_cast<float*>(t1.mean); a.tw_invstd = const_cast<float*>(t1.invstd);
                        a.wp = e->packed_u + e->pk_c1[b][i]; a.N = kBottleneck; a.asc = asc_n1(e, b, (int)i);
                        a.dst = bt; a.ldd = kBottleneck; a.dsum = bsum; a.dsq = bsq; a.dstride = kBottleneck;
                        constexpr int WBN = SMG_C1WS_BN;                     // 128: one workgroup per 64-row tile covers all 128 output columns (ws.cuh)
                        using WG_ = WsGeoT<np_of(fwd_op(0)), WBN>;
                        const int nM = ns * pl.HWp / 64, nN = kBottleneck / WBN;
                        a.tm = TileMap{nM, nN, 0};
                        BY(e, ESZ(e) * ns * pl.HW * (d.cin + kBottleneck));
                        ProfScope ps(e, cs, K_C1, 2.0 * ns * pl.HW * d.cin * kBottleneck);
                        const size_t smem = WG_::smem_bytes(d.cin);
                        static bool raised[64] = {};
                        if (!raised[e->device & 63]) {
                            (void)hipFuncSetAttribute((const void*)conv1x1_fwd_ws_kernel<0, WBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                            raised[e->device & 63] = true;
                        }
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_fwd_ws_kernel<0, WBN>), dim3(tile_grid(a.tm)), dim3(512), smem, cs, a);
                    } else
                    if (pl.HWp % 128 == 0 && wg128 >= small_wgs) run(CfgP128x128{});
                    else if (wg64 < small_wgs) run(CfgP32x64{});
                    else run(CfgP64x64{});
                }
                const BnTab t2 = bn_table(e, e->sb_tab[b][i], e->max_streams, s0, kBottleneck, P + d.n2.w, P + d.n2.b);
                if (e->generic3x3) bn_stat(cs, t2, ns, bsum, bsq, kBottleneck, 0, kBottleneck, pl.HW);
                if (!e->generic3x3) {
                    // norm2 + relu + conv2 (3x3, 128 -> 32) with an LDS-resident input halo (halo.cuh)
                    Halo3x3FwdArgs a;
                    a.src = bt; a.lds_ = kBottleneck; a.pl = pl; a.C = kBottleneck;
                    a.ssum = bsum; a.ssq = bsq; a.sstride = kBottleneck; a.gamma = P + d.n2.w; a.beta = P + d.n2.b; a.eps = kEps;
                    a.tw_mean = const_cast<float*>(t2.mean); a.tw_invstd = const_cast<float*>(t2.invstd);
                    a.dst = xs(b); a.ldd = Ct; a.dcoff = d.cin;
                    a.dsum = xsum; a.dsq = xsq; a.dstride = Ct;
                    a.wu = e->packed_u + e->pk_hf[b][i]; a.asc = asc_n2(e, b, (int)i);
                    BY(e, ESZ(e) * ns * pl.HW * (kBottleneck + kGrowth));
                    ProfScope ps(e, cs, K_C3, 2.0 * ns * pl.HW * 9 * kBottleneck * kGrowth);
                    TraceScope ts(cs, K_C3, dim3(banded_grid(halo_tile(pl, ns) == 16 ? ((pl.H + 15) / 16) * ((pl.W + 15) / 16) : ((pl.H + 7) / 8) * ((pl.W + 7) / 8), ns)));      // dev stamps: SMG_TRACE_KIND=2
                    if (halo_tile(pl, ns) == 8) {
                        // small planes: the wave-specialised form (halo.cuh; 17.1 -> 15.2 us per launch.  At TS = 16 it measures
                        // 67.8 -> 62.8 us serialised and nothing on the step - two forward chains already fill each other's gaps there)
                        a.tiles_x = (pl.W + 7) / 8; a.n_tiles = ((pl.H + 7) / 8) * a.tiles_x; a.streams = ns;
                        static bool raised8[64][3] = {};      // two (halo, weights) buffers: 51 KB as built, 76 KB in the -DSMG_SPLIT16=0 A/B build (three pieces) - past the 64 KB default
                        if (!raised8[e->device & 63][e->prec]) {
                            PREC_DISPATCH(e, if ((HaloFwdSGeo<8, PREC>::smem_bytes_ws(kBottleneck)) > 64 * 1024)
                                                 (void)hipFuncSetAttribute((const void*)conv3x3_halo_fwd_kernel<8, PREC, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                           (HaloFwdSGeo<8, PREC>::smem_bytes_ws(kBottleneck))));
                            raised8[e->device & 63][e->prec] = true;
                        }
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_fwd_kernel<8, PREC, false, true>), dim3(banded_grid(a.n_tiles, ns)), dim3(512),
                                           (HaloFwdSGeo<8, PREC>::smem_bytes_ws(kBottleneck)), cs, a));
                    } else
                    if (halo_tile(pl, ns) == 16) {
                        a.tiles_x = (pl.W + 15) / 16; a.n_tiles = ((pl.H + 15) / 16) * a.tiles_x; a.streams = ns;
                        if (pl.H % 16 || pl.W % 16) {      // tiles hang over the edge: the bounds-checked instantiation
                            PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_fwd_kernel<16, PREC, true>), dim3(banded_grid(a.n_tiles, ns)), dim3(256),
                                               (HaloFwdSGeo<16, PREC>::smem_bytes(kBottleneck)), cs, a));
                        } else if (SMG_FWD16_WS) {      // (dev A/B, -DSMG_FWD16_WS=1: the wave-specialised form on the 16 x 16 tiles too)
                            static bool raised16[64][3] = {};
                            if (!raised16[e->device & 63][e->prec]) {
                                PREC_DISPATCH(e, (void)hipFuncSetAttribute((const void*)conv3x3_halo_fwd_kernel<16, PREC, false, SMG_FWD16_WS != 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                           (HaloFwdSGeo<16, PREC>::smem_bytes_ws(kBottleneck))));
                                raised16[e->device & 63][e->prec] = true;
                            }
                            PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_fwd_kernel<16, PREC, false, SMG_FWD16_WS != 0>), dim3(banded_grid(a.n_tiles, ns)), dim3(512),
                                               (HaloFwdSGeo<16, PREC>::smem_bytes_ws(kBottleneck)), cs, a));
                        } else {
                            PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_fwd_kernel<16, PREC>), dim3(banded_grid(a.n_tiles, ns)), dim3(256),
                                               (HaloFwdSGeo<16, PREC>::smem_bytes(kBottleneck)), cs, a));
                        }
                    }
                } else {   // norm2 + relu + conv2 (3x3, 128 -> 32), appended to the block buffer (generic implicit GEMM)
                    auto run = [&](auto tag) {
                        using Cfg = decltype(tag);
                        FwdConvP<Cfg, F_THREE> p{};
                        p.src = bt; p.lds_ = kBottleneck; p.ps = pl; p.po = pl; p.K = kBottleneck; p.asc = asc_n2(e, b, (int)i);
                        p.bt = t2; p.fresh0 = kBottleneck; p.fsum = bsum; p.fsq = bsq; p.fstride = kBottleneck; p.eps = kEps;
                        p.tw_mean = const_cast<float*>(t2.mean); p.tw_invstd = const_cast<float*>(t2.invstd);
                        p.wp = e->packed_u + e->pk_g3f[b][i]; p.K8tot = 9 * kBottleneck / 8; p.N = kGrowth;
                        p.dst = xs(b); p.ldd = Ct; p.dcoff = d.cin;
                        p.dsum = xsum; p.dsq = xsq; p.dstride = Ct;
                        BY(e, ESZ(e) * ns * pl.HW * (kBottleneck + kGrowth));
                        launch_gemm(e, cs, p, dim3(ns * pl.HWp / Cfg::BM, 1), K_C3, 2.0 * ns * pl.HW * 9 * kBottleneck * kGrowth);
                    };
                    if (pl.HWp % 128 == 0) run(CfgP128x32{}); else run(CfgP64x32{});
                }
                if (b == 3 && i + 1 == T.layers[b].size()) {
                    // the last layer's 32 channels have no dense-layer or transition consumer to finish their table entries (norm5 reads
                    // the sums); the backward reads the table (backward.hip: use_tabs), so finish them here
                    const BnTab t5 = bn_table(e, e->sx_tab[b], e->max_streams, s0, Ct, nullptr, nullptr);
                    bn_stat(cs, t5, ns, xsum, xsq, Ct, Ct - kGrowth, kGrowth, pl.HW);
                }
            }
            if (b < 3 && on()) {   // transition: norm + relu + (avgpool2 commuted in front of) conv 1x1
                const Plane pn = e->p_blk[b + 1];
                const int Cn = kBlockCtot[b + 1];
                const BnTab tt = bn_table(e, e->sx_tab[b], e->max_streams, s0, Ct, P + T.tnorm[b].w, P + T.tnorm[b].b);
                auto run = [&](auto tag, auto ptag) {
                    using Cfg = MC<decltype(tag), decltype(ptag)::value>;
                    FwdConvP<Cfg, F_POOL, decltype(ptag)::value> p{};
                    p.src = xs(b); p.lds_ = Ct; p.ps = pl; p.po = pn; p.K = Ct; p.asc = asc_trans(e, b);
                    p.bt = tt; p.fresh0 = Ct - kGrowth; p.fsum = xsum; p.fsq = xsq; p.fstride = Ct; p.eps = kEps;     // the block's last layer
                    p.tw_mean = const_cast<float*>(tt.mean); p.tw_invstd = const_cast<float*>(tt.invstd);
                    p.wp = e->packed_u + e->pk_t[b]; p.K8tot = Ct / 8; p.N = Ct / 2;
                    p.dst = xs(b + 1); p.ldd = Cn; p.dcoff = 0;
                    p.dsum = st_off(fsum(e, e->st_X[b + 1]), Cn); p.dsq = st_off(fsq(e, e->st_X[b + 1]), Cn); p.dstride = Cn;
                    BY(e, ESZ(e) * ns * ((double)pl.HW * Ct + (double)pn.HW * (Ct / 2)));
                    launch_gemm(e, cs, p, dim3(ns * pn.HWp / Cfg::BM, (Ct / 2) / Cfg::BN), K_TRANS, 2.0 * ns * pn.HW * Ct * (Ct / 2));
                };
                PREC_DISPATCH(e, if (pn.HWp % 128 == 0) run(CfgP128x128{}, PTAG); else run(CfgP64x128{}, PTAG));
            }
        }
        return 0;
    };
    // (two chains pay once a chain has work to overlap: a single-sample pass - 2 streams of 160^2 - is bound by the host's launch
    //  rate, and two chains launch every kernel twice: 6.2 -> 5.1 ms for config 3's enveloping-then-sucking head on one chain; 5 streams
    //  of 456^2 want two: 30.1 against 28.9 ms on one)
    const bool two_chains = NS >= 2 && (int64_t)NS * e->p_blk[0].HW >= 200000;
    if (two_chains && !e->prof && !e->serialize) {
        const int h = NS / 2;
        HIP_OK(hipEventRecord(e->ev_misc, st));                 // packed weights + batch description are ready
        HIP_OK(hipStreamWaitEvent(e->side, e->ev_misc, 0));
        int n_units = 1 + 3;
        for (int b = 0; b < 4; ++b) n_units += (int)T.layers[b].size();
        for (int u = 0; u < n_units; ++u) {
            if (trunk_chain(0, h, st, u, u + 1)) return -5;
            if (trunk_chain(h, NS - h, e->side, u, u + 1)) return -5;
        }
        HIP_OK(hipEventRecord(e->ev_end, e->side));             // join before the head reads every stream's features
        HIP_OK(hipStreamWaitEvent(st, e->ev_end, 0));
    } else {
        if (trunk_chain(0, NS, st, 0, 1 << 30)) return -5;      // profiling: one chain, per-kernel times stay per layer
    }
    e->prof_stage = -1;
    const Plane p4 = e->p_blk[3];
    {   // norm5 + two-stream concat
        FeatArgs a;
        a.x4 = e->X[3]; a.p4 = p4; a.xsum = fsum(e, e->st_X[3]); a.xsq = fsq(e, e->st_X[3]);
        a.gamma = P + T.norm5.w; a.beta = P + T.norm5.b; a.eps = kEps;
        a.pair_a = e->d_pair_a; a.pair_b = e->d_pair_b; a.F = e->F;
        a.fsum = fsum(e, e->st_F); a.fsq = fsq(e, e->st_F); a.chunk = 64;
        ProfScope ps(e, st, K_OTHER, 0);
        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(feat_kernel<PREC>), dim3(2, NP, (p4.HW + 63) / 64), dim3(256), 0, st, a));
    }
    {   // head norm0 + relu + conv0 (1x1, 2048 -> 64)
        const BnTab th = bn_table(e, e->sf_tab, e->max_pairs, 0, 2 * kFeat, P + Hd.n0.w, P + Hd.n0.b);
        bn_stat(st, th, NP, fsum(e, e->st_F), fsq(e, e->st_F), 2 * kFeat, 0, 2 * kFeat, p4.HW);
        auto run = [&](auto tag, auto ptag) {
                using Cfg = decltype(tag);
                FwdConvP<Cfg, F_ONE, decltype(ptag)::value, true> p{};      // fp32 feature buffers in every mode
        p.src = e->F; p.lds_ = 2 * kFeat; p.ps = p4; p.po = p4; p.K = 2 * kFeat; p.asc = asc_head0(e);
        p.bt = th; p.fresh0 = 2 * kFeat; p.fsum = fsum(e, e->st_F); p.fsq = fsq(e, e->st_F); p.fstride = 2 * kFeat; p.eps = kEps;
        p.tw_mean = const_cast<float*>(th.mean); p.tw_invstd = const_cast<float*>(th.invstd);
        p.wp = e->packed_u + e->pk_head0; p.K8tot = 2 * kFeat / 8; p.N = kHeadMid;
        p.dst = e->H1; p.ldd = kHeadMid; p.dcoff = 0;
        p.dsum = fsum(e, e->st_H1); p.dsq = fsq(e, e->st_H1); p.dstride = kHeadMid;
        BY(e, 4.0 * NP * p4.HW * (2 * kFeat + kHeadMid));
        launch_gemm(e, st, p, dim3(NP * p4.HWp / Cfg::BM, 1), K_HEAD0, 2.0 * NP * p4.HW * 2 * kFeat * kHeadMid);
            };
            PREC_DISPATCH(e, if (p4.HWp % 128 == 0) run(CfgP128x64{}, PTAG); else run(CfgP64x64{}, PTAG));
    }
    {   // head norm1 + relu + conv1 (20x20 valid)
        ValueArgs a;
        a.h1 = e->H1; a.p4 = p4; a.hsum = fsum(e, e->st_H1); a.hsq = fsq(e, e->st_H1);
        a.gamma = P + Hd.n1.w; a.beta = P + Hd.n1.b; a.eps = kEps;
        a.w2p = e->packed_f + e->pk_head1; a.q = q_out; a.out_ch = e->head_out; a.OH = e->OH; a.OW = e->OW;
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(value_conv_kernel, dim3(NP * e->head_out * e->OH * e->OW), dim3(256), 0, st, a);
    }
    {   // BN running statistics, in the reference's update order - and, with or without an update, the reference's NaN propagation: a
        // non-finite batch statistic of a stream (pair) turns the Q values of every sample that uses it into NaN
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(bn_update_kernel, dim3(8, (unsigned)e->n_bnupd), dim3(256), 0, st,
                           e->d_bnupd + (trunk_id * 3 + head_id) * e->bnupd_stride,
                           e->fstat, e->fstat + e->fstat_span, net->bufs, net->nbt, e->d_seq_t, n_seq_t, e->d_seq_h, n_seq_h,
                           e->d_pair_a, e->d_pair_b, NP, e->head_out * e->OH * e->OW, q_out, NS, (n_seq_t || n_seq_h) ? 1 : 0);
    }
    HIP_OK(hipGetLastError());
    e->f_stem1 = B->heightmaps_dev != nullptr;
    e->bw_phase0_done = false;
    e->have_fwd = true; e->f_trunk = trunk_id; e->f_head = head_id; e->f_streams = NS; e->f_pairs = NP;
    return 0;
}

