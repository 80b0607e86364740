// plan.h - host-side description of the network the reference builds
// (code/models.py:301-358 reinforcement_net, :15-69 reactive_net; torchvision
// densenet121 per SURVEY.md Appendix A) and of the flat parameter/buffer arrays the
// engine works on.  The ORDER of entries is the reference module's state_dict()
// order, so index i <-> key i of a reference snapshot.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace smg {

constexpr int kGrowth = 32;
constexpr int kBottleneck = 128;              // bn_size * growth
constexpr int kBlockLayers[4] = {6, 12, 24, 16};
constexpr int kBlockCin[4] = {64, 128, 256, 512};
constexpr int kBlockCtot[4] = {256, 512, 1024, 1024};
constexpr int kFeat = 1024;
constexpr int kHeadMid = 64;
constexpr int kHeadKernel = 20;

struct BnRef {
    int64_t w = 0, b = 0;          // offsets into params / grads
    int64_t rm = 0, rv = 0;        // offsets into bufs
    int64_t nbt = 0;               // index into nbt
    int C = 0;
};
struct ConvRef {
    int64_t w = 0;
    int cout = 0, cin = 0, k = 1;
    int64_t count() const { return (int64_t)cout * cin * k * k; }
};
struct DenseLayerRef { BnRef n1; ConvRef c1; BnRef n2; ConvRef c2; int cin = 0; };
struct TrunkRef {
    ConvRef conv0; BnRef norm0;
    std::vector<DenseLayerRef> layers[4];
    BnRef tnorm[3]; ConvRef tconv[3];
    BnRef norm5;
    int64_t cls_w = 0, cls_b = 0;
    int64_t p_begin = 0, p_feat_end = 0, p_end = 0;
};
struct HeadRef { BnRef n0; ConvRef c0; BnRef n1; ConvRef c1; int64_t p_begin = 0, p_end = 0; };

struct LayoutEntry { std::string name; int kind; int64_t offset; int ndim; int64_t shape[4]; };

struct Layout {
    TrunkRef trunk[3];
    HeadRef head[3];
    std::vector<LayoutEntry> entries;
    int64_t n_params = 0, n_bufs = 0, n_nbt = 0;
    int head_out = 1;
};

inline Layout build_layout(int head_out) {
    Layout L;
    L.head_out = head_out;
    auto add = [&](const std::string& name, int kind, int64_t off, std::initializer_list<int64_t> shp) {
        LayoutEntry e; e.name = name; e.kind = kind; e.offset = off; e.ndim = (int)shp.size();
        int i = 0; for (auto s : shp) e.shape[i++] = s; for (; i < 4; ++i) e.shape[i] = 1;
        L.entries.push_back(e);
    };
    auto conv = [&](const std::string& name, int cout, int cin, int k) {
        ConvRef c; c.w = L.n_params; c.cout = cout; c.cin = cin; c.k = k;
        add(name + ".weight", 0, c.w, {cout, cin, k, k});
        L.n_params += c.count();
        return c;
    };
    auto bn = [&](const std::string& name, int C) {
        BnRef r; r.C = C;
        r.w = L.n_params; add(name + ".weight", 0, r.w, {C}); L.n_params += C;
        r.b = L.n_params; add(name + ".bias", 0, r.b, {C}); L.n_params += C;
        r.rm = L.n_bufs; add(name + ".running_mean", 1, r.rm, {C}); L.n_bufs += C;
        r.rv = L.n_bufs; add(name + ".running_var", 2, r.rv, {C}); L.n_bufs += C;
        r.nbt = L.n_nbt; add(name + ".num_batches_tracked", 3, r.nbt, {}); L.n_nbt += 1;
        return r;
    };
    const char* trunk_names[3] = {"suction_depth_trunk", "grasp_depth_trunk", "gs_depth_trunk"};
    for (int t = 0; t < 3; ++t) {
        TrunkRef& T = L.trunk[t];
        std::string f = std::string(trunk_names[t]) + ".features.";
        T.p_begin = L.n_params;
        T.conv0 = conv(f + "conv0", 64, 3, 7);
        T.norm0 = bn(f + "norm0", 64);
        int c = 64;
        for (int b = 0; b < 4; ++b) {
            for (int i = 0; i < kBlockLayers[b]; ++i) {
                std::string p = f + "denseblock" + std::to_string(b + 1) + ".denselayer" + std::to_string(i + 1) + ".";
                DenseLayerRef d; d.cin = c + i * kGrowth;
                d.n1 = bn(p + "norm1", d.cin);
                d.c1 = conv(p + "conv1", kBottleneck, d.cin, 1);
                d.n2 = bn(p + "norm2", kBottleneck);
                d.c2 = conv(p + "conv2", kGrowth, kBottleneck, 3);
                T.layers[b].push_back(d);
            }
            c += kBlockLayers[b] * kGrowth;
            if (b < 3) {
                std::string p = f + "transition" + std::to_string(b + 1) + ".";
                T.tnorm[b] = bn(p + "norm", c);
                T.tconv[b] = conv(p + "conv", c / 2, c, 1);
                c /= 2;
            }
        }
        T.norm5 = bn(f + "norm5", c);
        T.p_feat_end = L.n_params;
        std::string cl = std::string(trunk_names[t]) + ".classifier";
        T.cls_w = L.n_params; add(cl + ".weight", 0, T.cls_w, {1000, 1024}); L.n_params += 1000 * 1024;
        T.cls_b = L.n_params; add(cl + ".bias", 0, T.cls_b, {1000}); L.n_params += 1000;
        T.p_end = L.n_params;
    }
    // heads: suctionnet_val ('suction-val-*'), graspnet_val ('grasp-val-*'),
    // gsnet_val ('grasp-val-*' again - code/models.py:336-343)
    const char* head_names[3] = {"suctionnet_val", "graspnet_val", "gsnet_val"};
    const char* head_pref[3] = {"suction", "grasp", "grasp"};
    for (int h = 0; h < 3; ++h) {
        HeadRef& H = L.head[h];
        std::string p = std::string(head_names[h]) + "." + head_pref[h] + "-val-";
        H.p_begin = L.n_params;
        H.n0 = bn(p + "norm0", 2 * kFeat);
        H.c0 = conv(p + "conv0", kHeadMid, 2 * kFeat, 1);
        H.n1 = bn(p + "norm1", kHeadMid);
        H.c1 = conv(p + "conv1", head_out, kHeadMid, kHeadKernel);
        H.p_end = L.n_params;
    }
    return L;
}

}  // namespace smg
