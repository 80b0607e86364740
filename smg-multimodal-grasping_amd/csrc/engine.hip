// engine.hip - host side of the MI355X affordance engine: workspace ownership, the
// walk over DenseNet-121 (forward and backward) and the C ABI of include/smg_hip.h.
//
// Replaces, for one (trunk, head) of the reference model:
//   reinforcement_net.forward / reactive_net.forward   code/models.py:361-586, :72-296
//   loss.backward() of Trainer.backprop                code/trainer.py:350-351
//   torch.optim.Adam.step                              code/trainer.py:383
#include "engine.h"

static thread_local std::string g_err;
int smg_fail(int code, const std::string& msg) { g_err = msg; return code; }

// ------------------------------------------------------------------------------------
// creation
// ------------------------------------------------------------------------------------
static int engine_build(smg_engine* e) {
    const Layout& L = *e->L;
    const int NS = e->max_streams, NP = e->max_pairs;
    const int S = e->S;
    const int H1 = (S - 1) / 2 + 1, H2 = (H1 - 1) / 2 + 1;
    e->p_img = make_plane(S, S);
    e->p_stem = make_plane(H1, H1);
    int h = H2;
    for (int b = 0; b < 4; ++b) { e->p_blk[b] = make_plane(h, h); h /= 2; }
    e->OH = e->OW = e->p_blk[3].H - kHeadKernel + 1;
    if (e->OH < 1) return fail(-22, "input_size too small for the 20x20 value head");

    if (getenv("SMG_SERIALIZE")) e->serialize = true;        // profiling runs: serialised from the first launch (bench.py --serialize)
    const int cross = getenv("SMG_CROSSCHECK") ? atoi(getenv("SMG_CROSSCHECK")) : 0;
    e->generic3x3 = cross & 1; e->generic_c1 = cross & 2; e->generic_w1 = cross & 4;

    ALLOC(e->img4, (int64_t)NS * e->p_img.HWp * 4);
    ALLOC(e->stem, (int64_t)NS * e->p_stem.HWp * 64);
    ALLOC(e->DY0, (int64_t)NS * e->p_stem.HWp * 64);
    ALLOC(e->argmax, (int64_t)NS * e->p_blk[0].HWp * 64);
    int64_t bt_total = 0;
    for (int b = 0; b < 4; ++b) {
        ALLOC(e->X[b], (int64_t)NS * e->p_blk[b].HWp * kBlockCtot[b]);
        ALLOC(e->G[b], (int64_t)NS * e->p_blk[b].HWp * kBlockCtot[b]);
        for (int i = 0; i < kBlockLayers[b]; ++i) {
            e->bt_off[b].push_back(bt_total);
            bt_total += (int64_t)NS * e->p_blk[b].HWp * kBottleneck;
        }
    }
    ALLOC(e->Bt, bt_total);
    for (int k = 0; k < kRing; ++k) {   // ring: see kRing
        ALLOC(e->D2[k], (int64_t)NS * e->p_blk[0].HWp * kBottleneck);
        ALLOC(e->GS[k], (int64_t)NS * e->p_blk[0].HWp * kGrowth);
        ALLOC(e->D2S[k], (int64_t)NS * (e->p_blk[0].HWp / kScaleBlock));
        HIP_OK(hipEventCreateWithFlags(&e->ev_gs[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&e->ev_d2[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&e->ev_side[k], hipEventDisableTiming));
    }
    ALLOC(e->DY2, (int64_t)NS * e->p_blk[0].HWp * kBottleneck);
    HIP_OK(hipEventCreateWithFlags(&e->ev_misc, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&e->ev_end, hipEventDisableTiming));
    HIP_OK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    e->part_floats = (int64_t)24 << 20;   // partial weight-gradient tiles: 96 MB, or what the 3x3 launches of a full batch want
    for (int b = 0; b < 4; ++b) {
        const Plane& pl = e->p_blk[b];
        for (int ts : {halo_tile(pl), 8}) {
            const int th = 8;                                    // weight-gradient tiles are ts x 8 pixels
            const int nt = ((pl.H + th - 1) / th) * ((pl.W + ts - 1) / ts);
            const int tpw = w3_tiles_per_wg(nt, ts, NS, INT64_MAX, (double)ts / th);
            e->part_floats = std::max<int64_t>(e->part_floats, (int64_t)((nt + tpw - 1) / tpw) * NS * 9 * 32 * kBottleneck);
        }
    }
    ALLOC(e->part, e->part_floats);
    ALLOC(e->F, (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat);
    ALLOC(e->DF, (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat);
    ALLOC(e->H1, (int64_t)NP * e->p_blk[3].HWp * kHeadMid);
    ALLOC(e->DH1, (int64_t)NP * e->p_blk[3].HWp * kHeadMid);

    // statistic arenas: identical carving for forward (sum, sumsq) and backward (s1, s2)
    int64_t off = 0;
    auto carve = [&](int n, int stride) { StatArr s{off, stride}; off += (int64_t)n * stride; return s; };
    e->st_stem = carve(NS, 64);
    for (int b = 0; b < 4; ++b) e->st_X[b] = carve(NS, kBlockCtot[b]);
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < kBlockLayers[b]; ++i) e->st_Bt[b].push_back(carve(NS, kBottleneck));
    e->st_F = carve(NP, 2 * kFeat);
    e->st_H1 = carve(NP, kHeadMid);
    e->fstat_span = e->bstat_span = off;
    e->bs_stem = e->st_stem; e->bs_F = e->st_F; e->bs_H1 = e->st_H1;
    for (int b = 0; b < 4; ++b) { e->bs_X[b] = e->st_X[b]; e->bs_Bt[b] = e->st_Bt[b]; }
    if (2 * off > kStatRepStride) return fail(-22, "batch too large for the statistic arenas (kStatRepStride in gemm.cuh)");
    ALLOC(e->fstat, kFStatRep > 1 ? (int64_t)kFStatRep * kStatRepStride : 2 * off);
    ALLOC(e->bstat, (int64_t)kStatRep * kStatRepStride);      // kStatRep replicas (gemm.cuh: stat_get / stat_rep), 2 * off doubles used of each
    {   // dbeta / dgamma scratch (engine.h): per dense layer [beta cin | gamma cin], per transition [beta Cp | gamma Cp]
        int at = 0;
        for (int b = 0; b < 4; ++b)
            for (int i = 0; i < kBlockLayers[b]; ++i) { e->db_off[b][i] = at; at += 2 * (kBlockCin[b] + i * kGrowth); }
        for (int b = 0; b < 3; ++b) { e->db_toff[b] = at; at += 2 * kBlockCtot[b]; }
        e->db_total = at;
        ALLOC(e->dbscr, (int64_t)kDbRep * at);
        HIP_OK(hipMemset(e->dbscr, 0, (size_t)kDbRep * at * sizeof(float)));
        e->n_dbseg = 2 * (6 + 12 + 24 + 16 + 3);
        HIP_OK(hipMalloc((void**)&e->d_dbseg, (size_t)3 * e->n_dbseg * sizeof(DbSeg)));
        for (int t = 0; t < 3; ++t) {
            const TrunkRef& T = L.trunk[t];
            if ((int)T.layers[0].size() != kBlockLayers[0]) continue;      // trunk slot not used by this net
            std::vector<DbSeg> v;
            for (int b = 0; b < 4; ++b)
                for (int i = 0; i < kBlockLayers[b]; ++i) {
                    const DenseLayerRef& d = T.layers[b][i];
                    v.push_back(DbSeg{d.n1.b, e->db_off[b][i], d.cin});
                    v.push_back(DbSeg{d.n1.w, e->db_off[b][i] + d.cin, d.cin});
                }
            for (int b = 0; b < 3; ++b) {
                v.push_back(DbSeg{T.tnorm[b].b, e->db_toff[b], kBlockCtot[b]});
                v.push_back(DbSeg{T.tnorm[b].w, e->db_toff[b] + kBlockCtot[b], kBlockCtot[b]});
            }
            HIP_OK(hipMemcpy(e->d_dbseg + (size_t)t * e->n_dbseg, v.data(), v.size() * sizeof(DbSeg), hipMemcpyHostToDevice));
        }
    }
    {   // BN statistic tables: mean | invstd, [rows][C] each
        int64_t po = 0;
        auto carve_t = [&](int rows, int C) { int64_t at = po; po += (int64_t)2 * rows * C; return at; };
        for (int b = 0; b < 4; ++b) {
            e->sx_tab[b] = carve_t(NS, kBlockCtot[b]);
            for (int i = 0; i < kBlockLayers[b]; ++i) e->sb_tab[b][i] = carve_t(NS, kBottleneck);
        }
        e->sf_tab = carve_t(NP, 2 * kFeat);
        e->stab_floats = po;
        ALLOC(e->stab, po);
    }

    // packed weights + descriptor tables (one table per trunk, one per head)
    int64_t pku = 0, pkf = 0;
    // op0: operand kind of the pack in precision mode 0 (kSplitOp for the hot classes' operands, 0 otherwise).  Every pack is
    // preceded by one header unit (the kind-3 scale of the tensor, scale_kernel); the returned offset is the first data unit.
    auto add_units = [&](std::vector<PackDesc>& v, int64_t src, int cout, int cin, int mode, int K, int N, int op0 = 0) {
        pku += 1;
        PackDesc d{}; d.src = src; d.dst = pku; d.cout = cout; d.cin = cin; d.mode = mode; d.K8tot = K / 8; d.N = N; d.count = 0; d.op0 = op0;
        v.push_back(d); int64_t at = pku; pku += (int64_t)NPIECE * (K / 8) * N; return at;
    };
    auto add_f32 = [&](std::vector<PackDesc>& v, int64_t src, int cout, int cin, int mode, int64_t count) {
        PackDesc d{}; d.src = src; d.dst = pkf; d.cout = cout; d.cin = cin; d.mode = mode; d.count = (int)count;
        v.push_back(d); int64_t at = pkf; pkf += count; return at;
    };
    for (int t = 0; t < 3; ++t) {
        pku = pkf = 0;   // every trunk packs into the same region (only one trunk is active per forward)
        const TrunkRef& T = L.trunk[t];
        std::vector<PackDesc>& v = e->h_pack[t];
        e->pk_conv0 = add_units(v, T.conv0.w, 64, 3, PK_STEM, 224, 64);
        e->pk_conv0_1 = add_units(v, T.conv0.w, 64, 3, PK_STEM1, 64, 64);      // one-channel stem: weights summed over the 3 input channels
        for (int b = 0; b < 4; ++b) {
            if (t == 0) { e->pk_c1[b].clear(); e->pk_d1[b].clear(); e->pk_g3f[b].clear(); e->pk_g3d[b].clear(); e->pk_hf[b].clear(); e->pk_hd[b].clear(); }
            for (size_t i = 0; i < T.layers[b].size(); ++i) {
                const DenseLayerRef& d = T.layers[b][i];
                int64_t a1 = add_units(v, d.c1.w, kBottleneck, d.cin, PK_T1, d.cin, kBottleneck, kSplitOp);
                int64_t a1d = add_units(v, d.c1.w, kBottleneck, d.cin, PK_D1, kBottleneck, d.cin, kSplitOp);
                int64_t g2 = 0, g3 = 0;
                const int64_t hf = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_HF, 9 * kBottleneck, kGrowth, kSplitOp);
                const int64_t hd = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_HD, 9 * kGrowth, kBottleneck, kSplitOp);
                if (e->generic3x3) {
                    g2 = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_3F, 9 * kBottleneck, kGrowth, kSplitOp);      // (FwdConvP F_THREE: a forward policy)
                    g3 = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_3D, 9 * kGrowth, kBottleneck);
                }
                if (t == 0) {
                    e->pk_c1[b].push_back(a1); e->pk_d1[b].push_back(a1d);
                    e->pk_g3f[b].push_back(g2); e->pk_g3d[b].push_back(g3); e->pk_hf[b].push_back(hf); e->pk_hd[b].push_back(hd);
                }
            }
            if (b < 3) {
                e->pk_t[b] = add_units(v, T.tconv[b].w, T.tconv[b].cout, T.tconv[b].cin, PK_T1, T.tconv[b].cin, T.tconv[b].cout, kSplitOp);
                e->pk_td[b] = add_units(v, T.tconv[b].w, T.tconv[b].cout, T.tconv[b].cin, PK_D1, T.tconv[b].cout, T.tconv[b].cin);
            }
        }
    }
    const int64_t trunk_pku = pku, trunk_pkf = pkf;
    for (int hd = 0; hd < 3; ++hd) {
        pku = trunk_pku; pkf = trunk_pkf;
        const HeadRef& H = L.head[hd];
        std::vector<PackDesc>& v = e->h_pack_head[hd];
        e->pk_head0 = add_units(v, H.c0.w, kHeadMid, 2 * kFeat, PK_T1, 2 * kFeat, kHeadMid, kSplitOp);
        e->pk_hd0 = add_units(v, H.c0.w, kHeadMid, 2 * kFeat, PK_D1, kHeadMid, 2 * kFeat);
        e->pk_head1 = add_f32(v, H.c1.w, e->head_out, kHeadMid, PK_HEAD, H.c1.count());
    }
    e->packed_units = pku + 4096;        // slack: tiles wider than N over-read whole units behind the array (never stored)
    e->packed_floats = pkf;
    ALLOC(e->packed_u, e->packed_units);
    HIP_OK(hipMemset(e->packed_u, 0, (size_t)e->packed_units * sizeof(u32x4)));
    ALLOC(e->packed_f, pkf + 4);
    e->max_pack = (int)(e->h_pack[0].size() + e->h_pack_head[0].size());
    // Descriptor tables are static per (trunk, head): upload all nine once, so a forward
    // never has to wait on a host->device copy of them.
    e->pack_stride = e->max_pack;
    ALLOC(e->d_pack, 9 * e->pack_stride);
    for (int t = 0; t < 3; ++t)
        for (int hd = 0; hd < 3; ++hd) {
            std::vector<PackDesc> v = e->h_pack[t];
            v.insert(v.end(), e->h_pack_head[hd].begin(), e->h_pack_head[hd].end());
            HIP_OK(hipMemcpy(e->d_pack + (t * 3 + hd) * e->pack_stride, v.data(), v.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
        }
    {   // activation-scale descriptors (gamma / beta of every BN + ReLU operand of a matrix product), one table per (trunk, head)
        e->n_asc = 2 * kDenseLayers + 4;
        ALLOC(e->asc, 2 * e->n_asc);
        ALLOC(e->d_asc, 9 * e->n_asc);
        std::vector<float> ones(2 * e->n_asc, 1.f);
        HIP_OK(hipMemcpy(e->asc, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
        for (int t = 0; t < 3; ++t)
            for (int hd = 0; hd < 3; ++hd) {
                const TrunkRef& T = L.trunk[t];
                const HeadRef& Hd = L.head[hd];
                if ((int)T.layers[0].size() != kBlockLayers[0]) continue;      // trunk slot not used by this net
                std::vector<ActScaleDesc> v;
                for (int b = 0; b < 4; ++b)
                    for (size_t i = 0; i < T.layers[b].size(); ++i) {
                        v.push_back(ActScaleDesc{T.layers[b][i].n1.w, T.layers[b][i].n1.b, T.layers[b][i].cin});
                        v.push_back(ActScaleDesc{T.layers[b][i].n2.w, T.layers[b][i].n2.b, kBottleneck});
                    }
                for (int b = 0; b < 3; ++b) v.push_back(ActScaleDesc{T.tnorm[b].w, T.tnorm[b].b, kBlockCtot[b]});
                v.push_back(ActScaleDesc{Hd.n0.w, Hd.n0.b, 2 * kFeat});
                if ((int)v.size() != e->n_asc) return fail(-22, "activation-scale table size");
                HIP_OK(hipMemcpy(e->d_asc + (t * 3 + hd) * e->n_asc, v.data(), v.size() * sizeof(ActScaleDesc), hipMemcpyHostToDevice));
            }
        e->gamax_words = (int64_t)2 * kDenseLayers * NS * kAmaxRep;
        ALLOC(e->gamax, e->gamax_words);
        HIP_OK(hipMemset(e->gamax, 0, (size_t)e->gamax_words * sizeof(unsigned)));
    }
    e->bnupd_stride = 128;
    ALLOC(e->d_bnupd, 9 * e->bnupd_stride);
    for (int t = 0; t < 3; ++t)
        for (int hd = 0; hd < 3; ++hd) {
            const TrunkRef& T = L.trunk[t];
            const HeadRef& Hd = L.head[hd];
            std::vector<BnUpdDesc> v;   // in the reference's module order
            auto add = [&](const BnRef& r, const StatArr& s, int count, int head) {
                BnUpdDesc d; d.rm = r.rm; d.rv = r.rv; d.nbt = r.nbt; d.stat_off = s.off; d.stride = s.stride; d.coff = 0;
                d.C = r.C; d.count = count; d.head = head; v.push_back(d);
            };
            add(T.norm0, e->st_stem, e->p_stem.HW, 0);
            for (int b = 0; b < 4; ++b) {
                for (size_t i = 0; i < T.layers[b].size(); ++i) {
                    add(T.layers[b][i].n1, e->st_X[b], e->p_blk[b].HW, 0);
                    add(T.layers[b][i].n2, e->st_Bt[b][i], e->p_blk[b].HW, 0);
                }
                if (b < 3) add(T.tnorm[b], e->st_X[b], e->p_blk[b].HW, 0);
            }
            add(T.norm5, e->st_X[3], e->p_blk[3].HW, 0);
            add(Hd.n0, e->st_F, e->p_blk[3].HW, 1);
            add(Hd.n1, e->st_H1, e->p_blk[3].HW, 1);
            if ((int)v.size() > e->bnupd_stride) return fail(-22, "bn descriptor table overflow");
            e->n_bnupd = (int)v.size();
            HIP_OK(hipMemcpy(e->d_bnupd + (t * 3 + hd) * e->bnupd_stride, v.data(), v.size() * sizeof(BnUpdDesc), hipMemcpyHostToDevice));
        }

    // batch description: one device block, filled by one async copy from pinned memory
    const int R = NS > NP ? NS : NP;
    int so = 0;
    auto carve_i = [&](int n) { int at = so; so += (n + 3) / 4 * 4; return at; };
    e->so_image = carve_i(NS); e->so_rot = carve_i(NS); e->so_pa = carve_i(NP); e->so_pb = carve_i(NP);
    e->so_seq_t = carve_i(4 * R + 16); e->so_seq_h = carve_i(4 * R + 16);
    e->so_uptr = carve_i(NS + 1); e->so_upair = carve_i(2 * NP); e->so_uslot = carve_i(2 * NP);
    e->so_aff = carve_i(6 * NS); e->so_ma = carve_i(NS); e->so_mb = carve_i(NS);
    e->so_adam = carve_i(4);                  // graph path: {lr / (1 - beta1^t), sqrt(1 - beta2^t)} of the trunk range, of the head range
    e->stage_ints = so;
    ALLOC(e->d_stage, so);
    e->d_stream_image = e->d_stage + e->so_image; e->d_stream_rot = e->d_stage + e->so_rot;
    e->d_pair_a = e->d_stage + e->so_pa; e->d_pair_b = e->d_stage + e->so_pb;
    e->d_seq_t = e->d_stage + e->so_seq_t; e->d_seq_h = e->d_stage + e->so_seq_h;
    e->d_user_ptr = e->d_stage + e->so_uptr; e->d_user_pair = e->d_stage + e->so_upair; e->d_user_slot = e->d_stage + e->so_uslot;
    e->d_affine = (float*)(e->d_stage + e->so_aff);
    for (int k = 0; k < 2; ++k) {
        HIP_OK(hipHostMalloc((void**)&e->h_stage[k], (size_t)so * sizeof(int), hipHostMallocDefault));
        HIP_OK(hipEventCreateWithFlags(&e->ev_stage[k], hipEventDisableTiming));
    }
    HIP_OK(hipHostMalloc((void**)&e->h_stage_g, (size_t)so * sizeof(int), hipHostMallocDefault));
    memset(e->h_stage_g, 0, (size_t)so * sizeof(int));
    HIP_OK(hipEventCreateWithFlags(&e->ev_graph, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&e->ev_gin, hipEventDisableTiming));
    HIP_OK(hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking));
    return 0;
}

// ------------------------------------------------------------------------------------
// One training step as a replayable hipGraph
// ------------------------------------------------------------------------------------
// Everything a captured step bakes into its nodes: a call with another key is captured anew (the old graph is dropped).
struct StepKey {
    const void* params; const void* grads; const void* bufs; const void* nbt;
    int trunk_id, head_id, n_streams, n_pairs, n_images, hm_size, n_seq_t, n_seq_h, n_masks, loss_mode, prec, deterministic;
    const void* images; const void* heightmaps; const void* masks; double mean, stdv;
    const void* labels; const void* q; const void* loss; const void* dq; const void* m; const void* v; float lr, b1, b2, eps;
    hipStream_t stream;
    bool operator==(const StepKey& o) const { return memcmp(this, &o, sizeof(StepKey)) == 0; }
};
struct StepGraph {
    StepKey key; bool warm = false;           // warm: one eager step with this key has run (lazy function attributes, static tables)
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
};
static void step_graph_drop(smg_engine* e) {
    if (!e->step_graph) return;
    if (e->step_graph->exec) (void)hipGraphExecDestroy(e->step_graph->exec);
    if (e->step_graph->graph) (void)hipGraphDestroy(e->step_graph->graph);
    delete e->step_graph;
    e->step_graph = nullptr;
}
static void adam_scalars(const smg_adam* a, int step, float* sc) {
    const double bc1 = 1.0 - std::pow((double)a->beta1, step), bc2 = 1.0 - std::pow((double)a->beta2, step);
    sc[0] = (float)((double)a->lr / bc1);
    sc[1] = (float)std::sqrt(bc2);
}
// The launches of one step on `st`: zero the gradient ranges, forward, loss, backward, Adam (scalars from the device block).
static int step_body(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* B, int loss_mode, const float* labels,
                     float* q, float* loss, float* dq, const smg_adam* adam, hipStream_t st) {
    const Layout& L = *e->L;
    const int64_t t0 = L.trunk[trunk_id].p_begin, tn = L.trunk[trunk_id].p_feat_end - t0, h0 = L.head[head_id].p_begin, hn = L.head[head_id].p_end - h0;
    HIP_OK(hipMemsetAsync(net->grads + t0, 0, (size_t)tn * sizeof(float), st));
    HIP_OK(hipMemsetAsync(net->grads + h0, 0, (size_t)hn * sizeof(float), st));
    if (int rc = do_forward(e, net, trunk_id, head_id, B, q, st)) return rc;
    const int per_pair = e->head_out * e->OH * e->OW;
    hipLaunchKernelGGL(loss_kernel, dim3((B->n_pairs + 63) / 64), dim3(64), 0, st, loss_mode, (const float*)q, labels, B->n_pairs, per_pair, loss, dq);
    if (int rc = do_backward(e, net, dq, st)) return rc;
    const float* sc = reinterpret_cast<const float*>(e->d_stage + e->so_adam);
    const int64_t off[2] = {t0, h0}, cnt[2] = {tn, hn};
    for (int k = 0; k < 2; ++k) {
        int blocks = (int)((cnt[k] + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, st, net->params + off[k], (const float*)net->grads + off[k], adam->m + off[k], adam->v + off[k],
                           cnt[k], sc + 2 * k, adam->beta1, adam->beta2, adam->eps);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
extern "C" {

const char* smg_last_error(void) { return g_err.c_str(); }
int smg_version(void) { return SMG_ABI_VERSION; }
int smg_abi_struct_bytes(int which) {
    if (which == 0) return (int)sizeof(smg_batch);
    if (which == 1) return (int)sizeof(smg_net);
    if (which == 2) return (int)sizeof(smg_adam);
    return fail(-22, "smg_abi_struct_bytes: 0 = smg_batch, 1 = smg_net, 2 = smg_adam");
}

int smg_layout_count(int head_out) { return (int)layout_for(head_out).entries.size(); }
int64_t smg_layout_param_floats(int head_out) { return layout_for(head_out).n_params; }
int64_t smg_layout_buffer_floats(int head_out) { return layout_for(head_out).n_bufs; }
int64_t smg_layout_nbt_count(int head_out) { return layout_for(head_out).n_nbt; }
int smg_layout_entry(int head_out, int index, char* name, int name_cap, int* kind, int64_t* offset, int* ndim, int64_t shape[4]) {
    const Layout& L = layout_for(head_out);
    if (index < 0 || index >= (int)L.entries.size()) return fail(-22, "layout index out of range");
    const LayoutEntry& en = L.entries[index];
    if (name && name_cap > 0) { std::strncpy(name, en.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (kind) *kind = en.kind;
    if (offset) *offset = en.offset;
    if (ndim) *ndim = en.ndim;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = en.shape[i];
    return 0;
}
int smg_layout_trunk_range(int head_out, int trunk_id, int64_t* offset, int64_t* count) {
    if (trunk_id < 0 || trunk_id > 2) return fail(-22, "trunk_id");
    const TrunkRef& T = layout_for(head_out).trunk[trunk_id];
    *offset = T.p_begin; *count = T.p_feat_end - T.p_begin; return 0;
}
int smg_layout_head_range(int head_out, int head_id, int64_t* offset, int64_t* count) {
    if (head_id < 0 || head_id > 2) return fail(-22, "head_id");
    const HeadRef& H = layout_for(head_out).head[head_id];
    *offset = H.p_begin; *count = H.p_end - H.p_begin; return 0;
}

int smg_engine_create(int device, int input_size, int max_streams, int max_pairs, int head_out, smg_engine** out) {
    if (!out) return fail(-22, "out is NULL");
    if (head_out != 1 && head_out != 3) return fail(-22, "head_out must be 1 or 3");
    if (input_size < 640 || max_streams < 1 || max_pairs < 1) return fail(-22, "bad engine dimensions");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(-19, "no HIP device available: the affordance engine needs an MI355X");
    if (device < 0 || device >= ndev) return fail(-22, "device index out of range");
    HIP_OK(hipSetDevice(device));
    smg_engine* e = new smg_engine();
    e->device = device; e->S = input_size; e->max_streams = max_streams; e->max_pairs = max_pairs; e->head_out = head_out;
    e->L = &layout_for(head_out);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cu = cus;
    }
    int r = engine_build(e);
    if (r) { smg_engine_destroy(e); return r; }
    *out = e;
    return 0;
}

void smg_engine_destroy(smg_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {e->img4, e->stem, e->DY0, e->argmax, e->X[0], e->X[1], e->X[2], e->X[3], e->G[0], e->G[1], e->G[2], e->G[3],
                    e->Bt, e->DY2, e->part, e->F, e->DF, e->H1, e->DH1, e->fstat, e->bstat, e->dbscr, e->d_dbseg, e->asc, e->d_asc, e->gamax, e->packed_u, e->packed_f, e->stab, e->d_pack, e->d_bnupd,
                    e->d_stage};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (e->dbg_gsnap) (void)hipFree(e->dbg_gsnap);
    step_graph_drop(e);
    if (e->h_stage_g) (void)hipHostFree(e->h_stage_g);
    if (e->ev_graph) (void)hipEventDestroy(e->ev_graph);
    if (e->ev_gin) (void)hipEventDestroy(e->ev_gin);
    if (e->gstream) (void)hipStreamDestroy(e->gstream);
    for (int k = 0; k < 2; ++k) { if (e->h_stage[k]) (void)hipHostFree(e->h_stage[k]); if (e->ev_stage[k]) (void)hipEventDestroy(e->ev_stage[k]); }
    for (int k = 0; k < kRing; ++k) {
        if (e->D2[k]) (void)hipFree(e->D2[k]);
        if (e->GS[k]) (void)hipFree(e->GS[k]);
        if (e->D2S[k]) (void)hipFree(e->D2S[k]);
        if (e->ev_gs[k]) (void)hipEventDestroy(e->ev_gs[k]); if (e->ev_d2[k]) (void)hipEventDestroy(e->ev_d2[k]); if (e->ev_side[k]) (void)hipEventDestroy(e->ev_side[k]);
    }
    if (e->ev_misc) (void)hipEventDestroy(e->ev_misc);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    if (e->side) (void)hipStreamDestroy(e->side);
    for (auto& r : e->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    delete e;
}

int64_t smg_engine_workspace_bytes(const smg_engine* e) { return e ? e->workspace_bytes : 0; }

int smg_engine_geometry(const smg_engine* e, int H[6], int HWp[6]) {
    if (!e) return fail(-22, "engine is NULL");
    const Plane* pl[6] = {&e->p_img, &e->p_stem, &e->p_blk[0], &e->p_blk[1], &e->p_blk[2], &e->p_blk[3]};
    for (int i = 0; i < 6; ++i) { H[i] = pl[i]->H; HWp[i] = pl[i]->HWp; }
    return 0;
}

int smg_forward(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* batch, float* q_out_dev, void* stream) {
    if (!e || !net || !batch || !q_out_dev) return fail(-22, "NULL argument");
    if (!net->params || !net->bufs || !net->nbt) return fail(-22, "net arrays are NULL");
    if (trunk_id < 0 || trunk_id > 2 || head_id < 0 || head_id > 2) return fail(-22, "trunk_id / head_id out of range");
    HIP_OK(hipSetDevice(e->device));
    return do_forward(e, net, trunk_id, head_id, batch, q_out_dev, (hipStream_t)stream);
}

int smg_loss(smg_engine* e, int mode, const float* q_dev, const float* labels_dev, int n_pairs, float* loss_dev, float* dq_dev, void* stream) {
    if (!e || !q_dev || !labels_dev || !loss_dev || !dq_dev) return fail(-22, "NULL argument");
    if (mode == 1 && e->head_out != 3) return fail(-22, "cross-entropy loss needs a 3-class head");
    if (mode != 0 && mode != 1) return fail(-22, "loss mode must be 0 (Huber) or 1 (cross entropy)");
    if (n_pairs < 1 || n_pairs > e->max_pairs) return fail(-22, "n_pairs exceeds the engine's max_pairs");
    HIP_OK(hipSetDevice(e->device));
    const int per_pair = e->head_out * e->OH * e->OW;
    hipLaunchKernelGGL(loss_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, mode, q_dev, labels_dev, n_pairs, per_pair, loss_dev, dq_dev);
    HIP_OK(hipGetLastError());
    return 0;
}

int smg_backward(smg_engine* e, const smg_net* net, const float* dq_dev, void* stream) {
    if (!e || !net || !dq_dev) return fail(-22, "NULL argument");
    HIP_OK(hipSetDevice(e->device));
    return do_backward(e, net, dq_dev, (hipStream_t)stream);
}

int smg_backward_phase(smg_engine* e, const smg_net* net, const float* dq_dev, void* stream, int phase) {
    if (!e || !net || !dq_dev) return fail(-22, "NULL argument");
    if (phase != 0 && phase != 1) return fail(-22, "phase must be 0 or 1");
    HIP_OK(hipSetDevice(e->device));
    return do_backward(e, net, dq_dev, (hipStream_t)stream, phase == 0 ? 1 : 2);
}

int smg_layout_trunk_split(int head_out, int trunk_id, int64_t* offset) {
    if (trunk_id < 0 || trunk_id > 2 || !offset) return fail(-22, "trunk_id");
    *offset = layout_for(head_out).trunk[trunk_id].tnorm[0].w;       // first parameter behind dense block 1 (transition1.norm.weight)
    return 0;
}

int smg_train_step_graph(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* B, int loss_mode,
                         const float* labels_dev, float* q_out_dev, float* loss_out_dev, float* dq_dev, const smg_adam* adam, void* stream) {
    if (!e || !net || !B || !labels_dev || !q_out_dev || !loss_out_dev || !dq_dev || !adam) return fail(-22, "NULL argument");
    if (!net->params || !net->grads || !net->bufs || !net->nbt || !adam->m || !adam->v) return fail(-22, "net / optimizer arrays are NULL");
    if (trunk_id < 0 || trunk_id > 2 || head_id < 0 || head_id > 2) return fail(-22, "trunk_id / head_id out of range");
    if (loss_mode != 0 && loss_mode != 1) return fail(-22, "loss mode must be 0 (Huber) or 1 (cross entropy)");
    if (loss_mode == 1 && e->head_out != 3) return fail(-22, "cross-entropy loss needs a 3-class head");
    if (adam->step_trunk < 1 || adam->step_head < 1) return fail(-22, "Adam step counts are 1-based");
    if (e->prof || e->serialize) return fail(-22, "smg_train_step_graph: not while profiling / serialised (the graph spans two streams)");
    HIP_OK(hipSetDevice(e->device));
    if (int rc = validate_batch(e, B)) return rc;
    hipStream_t caller = (hipStream_t)stream, st = e->gstream;
    StepKey key;
    memset(&key, 0, sizeof(key));
    key.params = net->params; key.grads = net->grads; key.bufs = net->bufs; key.nbt = net->nbt;
    key.trunk_id = trunk_id; key.head_id = head_id; key.n_streams = B->n_streams; key.n_pairs = B->n_pairs; key.n_images = B->n_images;
    key.hm_size = B->hm_size; key.n_seq_t = B->bn_seq_trunk ? B->n_bn_seq_trunk : 0; key.n_seq_h = B->bn_seq_head ? B->n_bn_seq_head : 0;
    key.n_masks = B->masks_dev ? B->n_masks : 0; key.loss_mode = loss_mode; key.prec = e->prec; key.deterministic = e->deterministic;
    key.images = B->images_nchw_dev; key.heightmaps = B->heightmaps_dev; key.masks = B->masks_dev; key.mean = B->image_mean; key.stdv = B->image_std;
    key.labels = labels_dev; key.q = q_out_dev; key.loss = loss_out_dev; key.dq = dq_dev; key.m = adam->m; key.v = adam->v;
    key.lr = adam->lr; key.b1 = adam->beta1; key.b2 = adam->beta2; key.eps = adam->eps; key.stream = caller;
    HIP_OK(hipEventSynchronize(e->ev_graph));          // the previous step (it reads the pinned block; its graph may be dropped below)
    if (e->step_graph && !(e->step_graph->key == key)) step_graph_drop(e);
    if (!e->step_graph) { e->step_graph = new StepGraph(); e->step_graph->key = key; }
    StepGraph* g = e->step_graph;
    // the pinned block: the batch description and the Adam scalars of THIS step
    fill_stage(e, B, e->h_stage_g);
    float* sc = reinterpret_cast<float*>(e->h_stage_g + e->so_adam);
    adam_scalars(adam, adam->step_trunk, sc);
    adam_scalars(adam, adam->step_head, sc + 2);
    HIP_OK(hipEventRecord(e->ev_gin, caller));        // the step starts behind everything the caller has enqueued (input copies, label fill)
    HIP_OK(hipStreamWaitEvent(st, e->ev_gin, 0));
    e->capturing = true;           // (also for the eager first step: the same pinned block, the same launches)
    int rc = 0;
    if (!g->warm) {
        rc = step_body(e, net, trunk_id, head_id, B, loss_mode, labels_dev, q_out_dev, loss_out_dev, dq_dev, adam, st);
        g->warm = rc == 0;
    } else {
        if (!g->exec) {
            hipError_t err = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            if (err != hipSuccess) { e->capturing = false; return fail(-5, std::string("hipStreamBeginCapture: ") + hipGetErrorString(err)); }
            rc = step_body(e, net, trunk_id, head_id, B, loss_mode, labels_dev, q_out_dev, loss_out_dev, dq_dev, adam, st);
            err = hipStreamEndCapture(st, &g->graph);
            if (rc == 0 && err != hipSuccess) rc = fail(-5, std::string("hipStreamEndCapture: ") + hipGetErrorString(err));
            if (rc == 0) {
                err = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
                if (err != hipSuccess) rc = fail(-5, std::string("hipGraphInstantiate: ") + hipGetErrorString(err));
            }
            if (rc) { e->capturing = false; step_graph_drop(e); return rc; }
        }
        hipError_t err = hipGraphLaunch(g->exec, st);
        if (err != hipSuccess) rc = fail(-5, std::string("hipGraphLaunch: ") + hipGetErrorString(err));
        // what the step leaves behind on the host side, as the eager calls would
        e->have_fwd = true; e->bw_phase0_done = false; e->f_trunk = trunk_id; e->f_head = head_id; e->f_streams = B->n_streams; e->f_pairs = B->n_pairs;
        e->f_stem1 = B->heightmaps_dev != nullptr;
    }
    e->capturing = false;
    if (rc == 0) {
        HIP_OK(hipEventRecord(e->ev_graph, st));
        HIP_OK(hipStreamWaitEvent(caller, e->ev_graph, 0));      // ... and the caller's stream continues behind it (loss read-back)
    }
    return rc;
}

int smg_engine_set_precision(smg_engine* e, int precision) {
    if (!e) return fail(-22, "engine is NULL");
    if (precision < 0 || precision > 2) return fail(-22, "precision must be 0 (fp32 storage, fp32-class split products), 1 (bf16 storage) or 2 (fp16 activations, bf16 gradients)");
    if (precision && e->generic3x3) return fail(-22, "SMG_CROSSCHECK=1 (the generic implicit-GEMM 3x3 path) exists in the fp32-class mode only");
    e->prec = precision;
    step_graph_drop(e);
    e->have_fwd = false;        // activations saved by a forward of another precision are not backward-compatible
    e->bw_phase0_done = false;
    return 0;
}

int smg_engine_set_option(smg_engine* e, const char* name, int value) {
    if (!e || !name) return fail(-22, "NULL argument");
    const std::string s(name);
    if (s == "deterministic") { e->deterministic = value != 0; return 0; }
    if (s == "serialize") { e->serialize = value != 0; return 0; }
    if (s == "debug_stop") { e->dbg_stop = value; return 0; }
    return fail(-22, "unknown engine option '" + s + "'");
}

int smg_heightmap(const double* depth_img_dev, int h, int w, const double* intrinsics3x3, const double* cam_pose4x4,
                  const double* inv_homography3x3, int out_w, int out_h, double* out_dev, void* stream) {
    if (!depth_img_dev || !intrinsics3x3 || !cam_pose4x4 || !inv_homography3x3 || !out_dev || h < 1 || w < 1 || out_w < 1 || out_h < 1)
        return fail(-22, "bad heightmap arguments");
    hipPointerAttribute_t attr;
    HIP_OK(hipPointerGetAttributes(&attr, depth_img_dev));
    HIP_OK(hipSetDevice(attr.device));
    HeightmapArgs a;
    a.depth = depth_img_dev; a.h = h; a.w = w;
    a.fx = intrinsics3x3[0]; a.fy = intrinsics3x3[4]; a.cx = intrinsics3x3[2]; a.cy = intrinsics3x3[5];
    a.r20 = cam_pose4x4[8]; a.r21 = cam_pose4x4[9]; a.r22 = cam_pose4x4[10]; a.t2 = cam_pose4x4[11];
    for (int i = 0; i < 9; ++i) a.mi[i] = inv_homography3x3[i];
    a.out = out_dev; a.ow = out_w; a.oh = out_h;
    hipLaunchKernelGGL(heightmap_warp_kernel, dim3((out_w + 255) / 256, out_h), dim3(256), 0, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int smg_argmax(const float* values_dev, int n, int* idx_out_dev, float* val_out_dev, void* stream) {
    if (!values_dev || !idx_out_dev || !val_out_dev || n < 1) return fail(-22, "bad argmax arguments");
    hipPointerAttribute_t attr;
    HIP_OK(hipPointerGetAttributes(&attr, values_dev));
    HIP_OK(hipSetDevice(attr.device));
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, values_dev, n, idx_out_dev, val_out_dev);
    HIP_OK(hipGetLastError());
    return 0;
}

int smg_adam_step(float* params, const float* grads, float* m, float* v, int64_t offset, int64_t count, int step, float lr,
                  float beta1, float beta2, float eps, void* stream) {
    if (!params || !grads || !m || !v || count < 0 || step < 1) return fail(-22, "bad Adam arguments");
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)std::sqrt(bc2);
    if (count == 0) return 0;
    {   // no engine argument: launch on the device that owns the parameter array
        hipPointerAttribute_t attr;
        HIP_OK(hipPointerGetAttributes(&attr, params));
        HIP_OK(hipSetDevice(attr.device));
    }
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params + offset, grads + offset, m + offset, v + offset,
                       count, step_size, beta1, beta2, eps, 1.0f, bc2_sqrt);
    HIP_OK(hipGetLastError());
    return 0;
}

int64_t smg_debug_read(smg_engine* e, const char* name, float* host_out, int64_t cap, void* stream) {
    if (!e || !name) return fail(-22, "NULL argument");
    const int NS = e->max_streams, NP = e->max_pairs;
    const float* src = nullptr; int64_t n = 0;
    std::string s(name);
    if (s == "img") {
        src = e->img4; n = (int64_t)NS * e->p_img.HWp * 4;
        if (e->f_stem1 && host_out) {          // one-channel stem: the buffer holds [streams][HWp] floats; present it as the 4-channel image
            if (cap < n) return fail(-22, "img: buffer too small");
            HIP_OK(hipSetDevice(e->device));
            HIP_OK(hipStreamSynchronize((hipStream_t)stream));
            std::vector<float> one((size_t)(n / 4));
            HIP_OK(hipMemcpy(one.data(), src, one.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n / 4; ++i) { host_out[4 * i] = host_out[4 * i + 1] = host_out[4 * i + 2] = one[(size_t)i]; host_out[4 * i + 3] = 0.f; }
            return n;
        }
    }
    else if (s == "stem") { src = e->stem; n = (int64_t)NS * e->p_stem.HWp * 64; }
    else if (s == "dy0") { src = e->DY0; n = (int64_t)NS * e->p_stem.HWp * 64; }
    else if (s == "feat") { src = e->F; n = (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat; }
    else if (s == "h1") { src = e->H1; n = (int64_t)NP * e->p_blk[3].HWp * kHeadMid; }
    else if (s == "df") { src = e->DF; n = (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat; }
    else if (s == "dh1") { src = e->DH1; n = (int64_t)NP * e->p_blk[3].HWp * kHeadMid; }
    else if (s.size() == 2 && (s[0] == 'x' || s[0] == 'g') && s[1] >= '1' && s[1] <= '4') {
        const int b = s[1] - '1';
        src = s[0] == 'x' ? e->X[b] : e->G[b]; n = (int64_t)NS * e->p_blk[b].HWp * kBlockCtot[b];
    } else if (s.size() >= 4 && s.substr(0, 2) == "bt") {   // "bt<block>_<layer>" 1-based
        int b = 0, i = 0;
        if (std::sscanf(name, "bt%d_%d", &b, &i) != 2 || b < 1 || b > 4 || i < 1 || i > kBlockLayers[b - 1]) return fail(-22, "bad bt name");
        src = el(e, e->Bt, e->bt_off[b - 1][i - 1]); n = (int64_t)NS * e->p_blk[b - 1].HWp * kBottleneck;      // bt_off counts ELEMENTS of the mode
    } else if (s == "gsnap") {    // G' of the debug_stop layer's block in front of that layer's 1x1 data gradients (first f_streams streams)
        src = e->dbg_gsnap; n = e->dbg_gsnap_floats;
        if (!src) return fail(-22, "gsnap: no debug_stop backward has run");
    } else if (s == "dy2") {      // precision mode 0: the raw 3x3 data gradient of the dense layer the backward processed last, [streams][HWp of its block][128]
        src = e->DY2; n = (int64_t)NS * e->p_blk[0].HWp * kBottleneck;
    } else if (s.size() >= 5 && (s.substr(0, 3) == "gs_" || s.substr(0, 3) == "d2_")) {
        // "gs_<block>_<layer>" / "d2_<block>_<layer>" (1-based): the ring slot of that dense layer's finished gradients - valid until kRing
        // further layers have run (debug_stop).  d2 in precision mode 0: rebuilt from the units, (h + l) * the block's inverse scale.
        int b = 0, i = 0;
        if (std::sscanf(name + 3, "%d_%d", &b, &i) != 2 || b < 1 || b > 4 || i < 1 || i > kBlockLayers[b - 1]) return fail(-22, "bad ring buffer name");
        const int slot = ring_pos(b - 1, i - 1) % kRing;
        const Plane& pl = e->p_blk[b - 1];
        const bool gs = s[0] == 'g';
        const int64_t per_stream = (int64_t)pl.HWp * (gs ? kGrowth : kBottleneck);
        n = (int64_t)NS * per_stream;
        if (!host_out) return n;
        const int ns_out = (int)std::min<int64_t>(NS, cap / per_stream);      // a prefix of whole streams
        n = ns_out * per_stream;
        if (e->prec != 0) return fail(-22, "ring buffer debug reads exist in precision mode 0 only");
        HIP_OK(hipSetDevice(e->device));
        HIP_OK(hipStreamSynchronize((hipStream_t)stream));
        if (gs || kSplitOp != 3) {
            HIP_OK(hipMemcpy(host_out, gs ? e->GS[slot] : e->D2[slot], n * sizeof(float), hipMemcpyDeviceToHost));
            return n;
        }
        std::vector<unsigned short> u((size_t)ns_out * per_stream * 2);
        std::vector<float> inv((size_t)NS * (pl.HWp / kScaleBlock));
        HIP_OK(hipMemcpy(u.data(), e->D2[slot], u.size() * sizeof(unsigned short), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(inv.data(), e->D2S[slot], inv.size() * sizeof(float), hipMemcpyDeviceToHost));
        auto h2f = [](unsigned h) -> double {
            const int ex = (h >> 10) & 31, man = h & 1023;
            const double v = ex == 0 ? std::ldexp((double)man, -24) : (ex == 31 ? (man ? NAN : INFINITY) : std::ldexp((double)(man | 1024), ex - 25));
            return (h & 0x8000u) ? -v : v;
        };
        for (int sn = 0; sn < ns_out; ++sn)
            for (int k8 = 0; k8 < kD2K8; ++k8)
                for (int p = 0; p < pl.HWp; ++p) {
                    const size_t uh = ((((size_t)sn * 2 + 0) * kD2K8 + k8) * pl.HWp + p) * 8, ul = ((((size_t)sn * 2 + 1) * kD2K8 + k8) * pl.HWp + p) * 8;
                    const double iv = inv[(size_t)sn * (pl.HWp / kScaleBlock) + p / kScaleBlock];
                    for (int j = 0; j < 8; ++j)
                        host_out[((size_t)sn * pl.HWp + p) * kBottleneck + 8 * k8 + j] = (float)((h2f(u[uh + j]) + h2f(u[ul + j])) * iv);
                }
        return n;
    } else if (s == "asc") {      // the activation scales {s, 1 / s} of the last forward: dense layers (norm1, norm2 per layer), transitions, head norm0
        src = e->asc; n = 2 * e->n_asc;
    } else if (s.size() >= 6 && s.substr(0, 5) == "fs_bt") {   // "fs_bt<block>_<layer>": the fp64 forward sums of that bottleneck, raw doubles [sum | sumsq][streams][128] in the float buffer
        int b = 0, i = 0;
        if (std::sscanf(name, "fs_bt%d_%d", &b, &i) != 2 || b < 1 || b > 4 || i < 1 || i > kBlockLayers[b - 1]) return fail(-22, "bad fs_bt name");
        n = (int64_t)4 * NS * kBottleneck;                     // floats = 2 x doubles
        if (!host_out) return n;
        if (cap < n) return fail(-22, "fs_bt: buffer too small");
        HIP_OK(hipSetDevice(e->device));
        HIP_OK(hipStreamSynchronize((hipStream_t)stream));
        HIP_OK(hipMemcpy(host_out, fsum(e, e->st_Bt[b - 1][i - 1]), (size_t)NS * kBottleneck * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(host_out + 2 * NS * kBottleneck, fsq(e, e->st_Bt[b - 1][i - 1]), (size_t)NS * kBottleneck * sizeof(double), hipMemcpyDeviceToHost));
        return n;
    } else return fail(-22, "unknown debug buffer");
    if (!host_out) return n;
    if (cap < n) n = cap;
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    const bool typed = e->prec != 0 && (s[0] == 'x' || s[0] == 'g' || s.substr(0, 2) == "bt");   // 16-bit storage in modes 1 / 2
    if (!typed) {
        HIP_OK(hipMemcpy(host_out, src, n * sizeof(float), hipMemcpyDeviceToHost));
        return n;
    }
    std::vector<unsigned short> raw((size_t)n);
    HIP_OK(hipMemcpy(raw.data(), src, n * sizeof(unsigned short), hipMemcpyDeviceToHost));
    const bool f16 = e->prec == 2 && s[0] != 'g';               // activations of mode 2; gradients are bf16 in both 16-bit modes
    for (int64_t i = 0; i < n; ++i) {
        unsigned u;
        const unsigned h = raw[(size_t)i];
        if (!f16) u = h << 16;
        else {
            const unsigned sign = (h & 0x8000u) << 16, ex = (h >> 10) & 31u, man = h & 1023u;
            if (ex == 0) {
                float v = (float)man * (1.0f / 16777216.0f);        // subnormal: man * 2^-24
                std::memcpy(&u, &v, 4); u |= sign;
            } else if (ex == 31) u = sign | 0x7F800000u | (man << 13);
            else u = sign | ((ex + 112u) << 23) | (man << 13);
        }
        std::memcpy(&host_out[i], &u, 4);
    }
    return n;
}

int smg_profile_enable(smg_engine* e, int on) {
    if (!e) return fail(-22, "engine is NULL");
    e->prof = on != 0;
    for (auto& r : e->recs) { e->ev_pool.push_back(r.a); e->ev_pool.push_back(r.b); }
    e->recs.clear();
    for (int s = 0; s < 5; ++s)
        for (int k = 0; k < K_COUNT; ++k) { e->prof_ms[s][k] = 0; e->prof_n[s][k] = 0; e->prof_flops[s][k] = 0; e->prof_bytes[s][k] = 0; }
    return 0;
}

int smg_profile_kinds(void) { return K_COUNT; }
const char* smg_profile_kind_name(int kind) { return (kind >= 0 && kind < K_COUNT) ? kKindNames[kind] : ""; }

// Drains the recorded events (synchronises them) and returns the totals of one kind.
int smg_profile_read(smg_engine* e, int kind, double* ms, int64_t* launches, double* flops) {
    if (!e || kind < 0 || kind >= 5 * K_COUNT) return fail(-22, "bad profile query");
    for (auto& r : e->recs) {
        float t = 0.f;
        (void)hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            const int slots[2] = {0, r.stage + 1};
            for (int q = 0; q < (r.stage >= 0 ? 2 : 1); ++q) {
                const int s = slots[q];
                e->prof_ms[s][r.kind] += t; e->prof_n[s][r.kind] += 1; e->prof_flops[s][r.kind] += r.flops; e->prof_bytes[s][r.kind] += r.bytes;
            }
        }
        e->ev_pool.push_back(r.a); e->ev_pool.push_back(r.b);
    }
    e->recs.clear();
    const int slot = kind / K_COUNT; kind %= K_COUNT;
    if (ms) *ms = e->prof_ms[slot][kind];
    if (launches) *launches = e->prof_n[slot][kind];
    if (flops) *flops = e->prof_flops[slot][kind];
    return 0;
}

// Algorithmic HBM bytes (see BY()) accumulated for one class since smg_profile_enable; call after smg_profile_read.
int smg_profile_read_bytes(smg_engine* e, int kind, double* bytes) {
    if (!e || kind < 0 || kind >= 5 * K_COUNT || !bytes) return fail(-22, "bad profile query");
    *bytes = e->prof_bytes[kind / K_COUNT][kind % K_COUNT];
    return 0;
}

}  // extern "C"
