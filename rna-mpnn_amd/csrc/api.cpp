// C-ABI layer of librnampnn_hip.so: handle, weight registry, workspace carving and the launch
// sequence of one forward pass.  See include/rnampnn_hip.h for the contract of every symbol.
#include "../../include/rnampnn_hip.h"
#include "rnampnn_internal.h"
#include "kernels_bf16.h"
#include "kernels_train.h"

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) return fail(RNAMPNN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

extern "C" const char* rnampnn_last_error(void) { return g_err; }
extern "C" const char* rnampnn_version(void) { return "rnampnn-hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------
struct RawT {                 // one state_dict entry in the reference's layout
    std::string key;
    int64_t numel;
    size_t off;               // float offset in the raw arena
    bool set;
};

struct Lin {                  // nn.Linear
    int in, out, in_pad;
    int w, b;                 // RawT indices
    size_t wt;                // derived: K-major f32 [in_pad][out]
    size_t wb;                // derived: bf16 [out][in_pad] (fast path), or (size_t)-1
    bool gelu;
};
struct Chain {                // fused FFN chain of the fast path (kernels_bf16.hip: k_ffn_chain)
    bool ok = false;
    int K0 = 0, H = 0, NH = 0, NOUT = 0, n_valid = 0;
    size_t img = 0;           // derived: fragment image of all layers
    size_t last_bias = 0;     // derived: bias of the last Linear padded to NOUT
};
struct Attn { Lin qkv, out; int gn_scale, gn_shift; size_t img_qkv = 0, img_out = 0; };    // img_*: fragment images of the fused per-RNA layer kernel
struct Bert { std::vector<Attn> attn; std::vector<Lin> ffn; int heads; Chain chain; };
struct Mlp2 {                 // message_layers / edge_layers of one ResMPNN
    int depth;
    int w[2], b[2];           // RawT indices
    size_t pq_t, pq_b;        // derived f32: [128][256] K-major (P | Q parts of Linear 0), bias [b1 | 0]
    size_t wc_t, w2_t;        // derived f32: e-part of Linear 0 and Linear 1, K-major
    size_t pq_img;            // derived bf16 [P | Q] fragment image for the fused node-update kernel
    size_t pq_bp;             // derived f32 bias of Linear 0 in the order of that image's P rows
    size_t img;               // derived bf16 fragment image of (Wc, W2) for the fused edge kernel
    size_t b2p;               // derived f32 bias of Linear 1 in the kernel's channel order
};
struct MpnnLayer { int gn_scale, gn_shift; Mlp2 msg, edge; };

struct rnampnn_ctx {
    RnaMpnnConfig cfg;
    std::vector<RawT> raw;
    size_t raw_floats = 0;
    float* raw_arena = nullptr;
    size_t der_bytes = 0;
    char* der_arena = nullptr;
    bool finalized = false;
    // structure
    Lin raw_project;
    Bert emb, post;
    int feat_gn_scale, feat_gn_shift;
    std::vector<Lin> edge_embed;
    size_t edge_embed_img = 0;    // bf16 fragment image for the fast path
    size_t edge_embed_b1p = 0;    // bias of its second Linear in the kernel's channel order
    std::vector<MpnnLayer> mpnn;
    std::vector<Lin> raw_ffn;
    Chain raw_chain;
    int rawffn_gn_scale, rawffn_gn_shift;
    std::vector<Lin> readout;
    Chain readout_chain;
    int fmax = 0;                 // widest node activation
    // optional live timing of the dominant kernel (bench.py roofline leg)
    bool prof = false;
    int prof_stride = 1;          // time every prof_stride-th fused launch (events cost ~6 us of stream idle each)
    long long prof_seen = 0;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<unsigned char> ev_kind;   // per event pair: 0 = <edge update, message> launch, 1 = message-only / edge-only launch
    double prof_ms = 0.0, prof_ms_kind[2] = {0.0, 0.0};
    long long prof_n = 0, prof_n_kind[2] = {0, 0};
    // tapes of rnampnn_train_forward calls whose backward may still come (the activations themselves live in the caller's
    // workspaces): one record per workspace, identified by a monotonically increasing id that rnampnn_train_backward must present
    struct Tape { int64_t id; int B, T, tnorm; float p; uint64_t seed; const void* ws; bool mixed; const unsigned long long* seed_dev;
                  bool att_mfma; };     // att_mfma: which attention kernels wrote the (m, l) statistics of this tape - the backward must recompute S with the same ones
    std::vector<Tape> tapes;
    int64_t tape_counter = 0;
    // optional: events the backward records on its stream once a chunk of the flat gradient is final (rnampnn_grad_chunks),
    // so that the caller's all-reduce of that chunk can run on a side stream under the rest of the backward
    hipEvent_t grad_ev[2] = {nullptr, nullptr};
    const unsigned long long* seed_dev = nullptr;   // rnampnn_set_seed_source: the training kernels read the dropout seed from device memory
    WImageCache* wimg = nullptr;   // prebuilt weight-fragment images of the bf16-mixed trainer (kernels_train.h)
    bool raw_external = false;     // raw_arena is the caller's flat parameter buffer (rnampnn_use_weight_arena)
};


static int add_raw(rnampnn_ctx* c, const std::string& key, int64_t numel) {
    RawT t{key, numel, c->raw_floats, false};
    c->raw_floats += (size_t)((numel + 3) / 4 * 4);
    c->raw.push_back(t);
    return (int)c->raw.size() - 1;
}
static size_t add_der(rnampnn_ctx* c, size_t bytes) {
    size_t off = c->der_bytes;
    c->der_bytes += (bytes + 255) / 256 * 256;
    return off;
}
static Lin make_lin(rnampnn_ctx* c, const std::string& prefix, int in, int out, bool gelu) {
    Lin l;
    l.in = in; l.out = out; l.in_pad = (in + 31) / 32 * 32; l.gelu = gelu;
    l.w = add_raw(c, prefix + ".weight", (int64_t)in * out);
    l.b = add_raw(c, prefix + ".bias", out);
    l.wt = add_der(c, (size_t)l.in_pad * out * sizeof(float));
    l.wb = add_der(c, (size_t)l.in_pad * out * sizeof(bf16_t));
    if (out > c->fmax) c->fmax = out;
    return l;
}
static void make_gn(rnampnn_ctx* c, const std::string& prefix, int& scale, int& shift) {
    scale = add_raw(c, prefix + ".scale", RN_D);
    shift = add_raw(c, prefix + ".shift", RN_D);
}
static Chain make_chain(rnampnn_ctx* c, const std::vector<Lin>& layers);
static Bert make_bert(rnampnn_ctx* c, const std::string& prefix, int n_attn, int heads, int ffn_dim, int n_ffn) {
    Bert b;
    b.heads = heads;
    for (int j = 0; j < n_attn; ++j) {              // registration order = torch state_dict order
        Attn a;
        std::string p = prefix + ".bi_attention_layers." + std::to_string(j);
        a.qkv.in = RN_D; a.qkv.out = 3 * RN_D; a.qkv.in_pad = RN_D; a.qkv.gelu = false;
        a.qkv.w = add_raw(c, p + ".in_proj_weight", 3 * RN_D * RN_D);
        a.qkv.b = add_raw(c, p + ".in_proj_bias", 3 * RN_D);
        a.qkv.wt = add_der(c, (size_t)RN_D * 3 * RN_D * sizeof(float));
        a.qkv.wb = add_der(c, (size_t)RN_D * 3 * RN_D * sizeof(bf16_t));
        a.out = make_lin(c, p + ".out_proj", RN_D, RN_D, false);
        a.img_qkv = add_der(c, (size_t)12 * 8 * 1024);
        a.img_out = add_der(c, (size_t)4 * 8 * 1024);
        b.attn.push_back(a);
    }
    if (3 * RN_D > c->fmax && n_attn > 0) c->fmax = 3 * RN_D;
    for (int j = 0; j < n_attn; ++j)
        make_gn(c, prefix + ".graph_norm_layers." + std::to_string(j), b.attn[j].gn_scale, b.attn[j].gn_shift);
    int in = RN_D;
    for (int i = 0; i < n_ffn; ++i) {
        b.ffn.push_back(make_lin(c, prefix + ".ffn_layers." + std::to_string(3 * i), in, ffn_dim, true));
        in = ffn_dim;
    }
    b.ffn.push_back(make_lin(c, prefix + ".ffn_layers." + std::to_string(3 * n_ffn), ffn_dim, RN_D, false));
    b.chain = make_chain(c, b.ffn);
    return b;
}
// layers = [K0->H gelu] + NH x [H->H gelu] + [H->n_out plain]; fused when a kernel instance exists
static Chain make_chain(rnampnn_ctx* c, const std::vector<Lin>& layers) {
    Chain ch;
    if (c->cfg.precision != RNAMPNN_PREC_BF16 || layers.size() < 2 || layers.size() > 4) return ch;
    const Lin& first = layers.front();
    const Lin& last = layers.back();
    int H = first.out;
    for (size_t i = 1; i + 1 < layers.size(); ++i)
        if (layers[i].in != H || layers[i].out != H) return ch;
    if (last.in != H || !first.gelu || last.gelu) return ch;
    ch.K0 = first.in_pad; ch.H = H; ch.NH = (int)layers.size() - 2; ch.NOUT = (last.out + 31) / 32 * 32; ch.n_valid = last.out;
    bool have = (ch.K0 == 128 && H == 512 && ch.NH == 2 && ch.NOUT == 128) || (ch.K0 == 32 && H == 512 && ch.NH == 2 && ch.NOUT == 128) ||
                (ch.K0 == 256 && H == 512 && ch.NH == 0 && ch.NOUT == 32);
    if (!have) return ch;
    ch.ok = true;
    ch.img = add_der(c, chain_image_bytes(ch.K0, ch.H, ch.NH, ch.NOUT));
    ch.last_bias = add_der(c, ch.NOUT * sizeof(float));
    return ch;
}

static Mlp2 make_mlp2(rnampnn_ctx* c, const std::string& prefix, int depth) {
    Mlp2 m;
    m.depth = depth;
    m.w[1] = m.b[1] = -1;
    for (int i = 0; i < depth; ++i) {
        std::string p = prefix + "." + std::to_string(3 * i);
        m.w[i] = add_raw(c, p + ".weight", (int64_t)RN_D * (i == 0 ? 3 * RN_D : RN_D));
        m.b[i] = add_raw(c, p + ".bias", RN_D);
    }
    m.pq_t = add_der(c, (size_t)RN_D * 256 * sizeof(float));
    m.pq_b = add_der(c, 256 * sizeof(float));
    m.wc_t = add_der(c, (size_t)RN_D * RN_D * sizeof(float));
    m.w2_t = add_der(c, (size_t)RN_D * RN_D * sizeof(float));
    m.pq_img = add_der(c, (size_t)64 * 1024);
    m.pq_bp = add_der(c, RN_D * sizeof(float));
    m.img = add_der(c, (size_t)2 * RN_D * RN_D * sizeof(bf16_t));
    m.b2p = add_der(c, RN_D * sizeof(float));
    return m;
}

extern "C" int rnampnn_create(const RnaMpnnConfig* cfg, rnampnn_handle* out) {
    if (!cfg || !out) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_create: null argument");
    const RnaMpnnConfig& g = *cfg;
    if (g.res_embedding_dim != RN_D || g.res_edge_embedding_dim != RN_D || g.raw_embedding_dim != RN_D)
        return fail(RNAMPNN_ERR_UNSUPPORTED, "embedding dims must be 128 (got %d/%d/%d)", g.res_embedding_dim,
                    g.res_edge_embedding_dim, g.raw_embedding_dim);
    if (g.num_res_neighbours < 1) return fail(RNAMPNN_ERR_BAD_ARG, "num_res_neighbours must be >= 1");
    if (g.num_res_neighbours > RNAMPNN_KMAX)
        return fail(RNAMPNN_ERR_K_TOO_LARGE, "num_res_neighbours %d > %d", g.num_res_neighbours, RNAMPNN_KMAX);
    auto depth_ok = [](int d) { return d >= 1 && d <= 2; };
    if (!depth_ok(g.depth_res_edge_feature) || !depth_ok(g.depth_res_mpnn) || !depth_ok(g.num_mpnn_edge_layers))
        return fail(RNAMPNN_ERR_UNSUPPORTED, "edge / message MLP depth must be 1 or 2");
    auto dim_ok = [](int d) { return d >= 32 && d <= 2048 && d % 32 == 0; };
    if (!dim_ok(g.embedding_ffn_dim) || !dim_ok(g.post_fusion_ffn_dim) || !dim_ok(g.num_raw_ffn_dim) ||
        !dim_ok(g.readout_hidden_dim))
        return fail(RNAMPNN_ERR_UNSUPPORTED, "FFN widths must be multiples of 32 in [32, 2048]");
    if (g.num_embedding_ffn_layers < 1 || g.num_post_fusion_ffn_layers < 1 || g.num_raw_ffn_layers < 1 ||
        g.num_readout_layers < 1 || g.num_res_mpnn_layers < 1)
        return fail(RNAMPNN_ERR_UNSUPPORTED, "layer counts must be >= 1");
    auto heads_ok = [](int n_attn, int h) { return n_attn == 0 || h == 2 || h == 4 || h == 8 || h == 16; };
    if (g.num_embedding_attn_layers < 0 || g.num_post_fusion_attn_layers < 0 ||
        !heads_ok(g.num_embedding_attn_layers, g.num_embedding_heads) ||
        !heads_ok(g.num_post_fusion_attn_layers, g.num_post_fusion_heads))
        return fail(RNAMPNN_ERR_UNSUPPORTED, "attention heads must be 2, 4, 8 or 16");
    if (g.padding_len < 1) return fail(RNAMPNN_ERR_BAD_ARG, "padding_len must be positive");
    if (g.precision != RNAMPNN_PREC_F32 && g.precision != RNAMPNN_PREC_BF16)
        return fail(RNAMPNN_ERR_BAD_ARG, "unknown precision %d", g.precision);
    if (g.precision == RNAMPNN_PREC_BF16 && (g.depth_res_edge_feature != 2 || g.depth_res_mpnn != 2))
        return fail(RNAMPNN_ERR_UNSUPPORTED, "the bf16 kernels cover depth-2 edge-embedding / message MLPs (edge-update MLP: depth 1 or 2); "
                    "use precision f32");

    rnampnn_ctx* c = new rnampnn_ctx();
    c->cfg = g;
    c->fmax = 3 * RN_D;
    // registration order follows torch's state_dict order of RNAMPNN (rnampnn.py:94-134)
    c->raw_project = make_lin(c, "res_feature.raw_project", RN_RAW, RN_D, false);
    c->emb = make_bert(c, "res_feature.res_embedding", g.num_embedding_attn_layers, g.num_embedding_heads,
                       g.embedding_ffn_dim, g.num_embedding_ffn_layers);
    make_gn(c, "res_feature.graph_norm", c->feat_gn_scale, c->feat_gn_shift);
    for (int i = 0; i < g.depth_res_edge_feature; ++i)
        c->edge_embed.push_back(make_lin(c, "res_feature.res_edge_embedding_layers." + std::to_string(3 * i),
                                         i == 0 ? RN_ERAW : RN_D, RN_D, true));
    c->edge_embed_img = add_der(c, (size_t)(4 * 7 + 32) * 1024);
    c->edge_embed_b1p = add_der(c, RN_D * sizeof(float));
    for (int l = 0; l < g.num_res_mpnn_layers; ++l) {
        MpnnLayer m;
        std::string p = "res_mpnn_layers." + std::to_string(l);
        make_gn(c, p + ".graph_norm", m.gn_scale, m.gn_shift);
        m.msg = make_mlp2(c, p + ".message_layers", g.depth_res_mpnn);
        m.edge = make_mlp2(c, p + ".edge_layers", g.num_mpnn_edge_layers);
        c->mpnn.push_back(m);
    }
    c->post = make_bert(c, "post_fusion", g.num_post_fusion_attn_layers, g.num_post_fusion_heads,
                        g.post_fusion_ffn_dim, g.num_post_fusion_ffn_layers);
    {
        int in = RN_RAW;
        for (int i = 0; i < g.num_raw_ffn_layers; ++i) {
            c->raw_ffn.push_back(make_lin(c, "raw_embedding.raw_ffn." + std::to_string(3 * i), in, g.num_raw_ffn_dim, true));
            in = g.num_raw_ffn_dim;
        }
        c->raw_ffn.push_back(make_lin(c, "raw_embedding.raw_ffn." + std::to_string(3 * g.num_raw_ffn_layers),
                                      g.num_raw_ffn_dim, RN_D, false));
        make_gn(c, "raw_embedding.graph_norm", c->rawffn_gn_scale, c->rawffn_gn_shift);
        c->raw_chain = make_chain(c, c->raw_ffn);
    }
    {
        int in = 2 * RN_D;
        for (int i = 0; i < g.num_readout_layers - 1; ++i) {
            c->readout.push_back(make_lin(c, "readout.readout_layers." + std::to_string(3 * i), in, g.readout_hidden_dim, true));
            in = g.readout_hidden_dim;
        }
        c->readout.push_back(make_lin(c, "readout.readout_layers." + std::to_string(3 * (g.num_readout_layers - 1)),
                                      in, 4, false));
        c->readout_chain = make_chain(c, c->readout);
    }
    *out = c;
    return RNAMPNN_OK;
}

extern "C" int rnampnn_profile_enable(rnampnn_handle h, int32_t enable) {
    if (!h) return fail(RNAMPNN_ERR_BAD_ARG, "null handle");
    if (enable && h->ev.empty()) {
        h->ev.resize(2 * 8192);
        h->ev_kind.assign(8192, 0);
        for (auto& e : h->ev) HIP_TRY(hipEventCreate(&e));
    }
    h->prof = enable != 0;
    h->prof_stride = enable > 1 ? enable : 1;
    h->prof_seen = 0;
    return RNAMPNN_OK;
}

static int profile_collect(rnampnn_handle h) {
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(h->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        const int kind = h->ev_kind[i / 2] ? 1 : 0;
        h->prof_ms += ms; h->prof_n += 1;
        h->prof_ms_kind[kind] += ms; h->prof_n_kind[kind] += 1;
    }
    h->ev_used = 0;
    return RNAMPNN_OK;
}
static void profile_reset(rnampnn_handle h) {
    h->prof_ms = 0.0; h->prof_n = 0;
    for (int i = 0; i < 2; ++i) { h->prof_ms_kind[i] = 0.0; h->prof_n_kind[i] = 0; }
}

extern "C" int rnampnn_profile_read(rnampnn_handle h, double* kernel_ms, int64_t* launches, int32_t reset) {
    if (!h) return fail(RNAMPNN_ERR_BAD_ARG, "null handle");
    int rc = profile_collect(h);
    if (rc) return rc;
    if (kernel_ms) *kernel_ms = h->prof_ms;
    if (launches) *launches = h->prof_n;
    if (reset) profile_reset(h);
    return RNAMPNN_OK;
}

extern "C" int rnampnn_profile_read_kinds(rnampnn_handle h, double* kernel_ms2, int64_t* launches2, int32_t reset) {
    if (!h || !kernel_ms2 || !launches2) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_profile_read_kinds: null argument");
    int rc = profile_collect(h);
    if (rc) return rc;
    for (int i = 0; i < 2; ++i) { kernel_ms2[i] = h->prof_ms_kind[i]; launches2[i] = h->prof_n_kind[i]; }
    if (reset) profile_reset(h);
    return RNAMPNN_OK;
}

extern "C" int rnampnn_destroy(rnampnn_handle h) {
    if (!h) return RNAMPNN_OK;
    for (auto& e : h->ev) (void)hipEventDestroy(e);
    if (h->raw_arena && !h->raw_external) (void)hipFree(h->raw_arena);
    if (h->der_arena) (void)hipFree(h->der_arena);
    t_wimg_destroy(h->wimg);
    delete h;
    return RNAMPNN_OK;
}

extern "C" int rnampnn_num_weights(rnampnn_handle h) { return h ? (int)h->raw.size() : 0; }

extern "C" int rnampnn_weight_info(rnampnn_handle h, int32_t i, const char** key, int64_t* numel) {
    if (!h || i < 0 || i >= (int)h->raw.size()) return fail(RNAMPNN_ERR_BAD_ARG, "weight index out of range");
    if (key) *key = h->raw[i].key.c_str();
    if (numel) *numel = h->raw[i].numel;
    return RNAMPNN_OK;
}

extern "C" int rnampnn_set_weight(rnampnn_handle h, const char* key, const float* data, int64_t numel,
                                  int32_t is_host, void* stream) {
    if (!h || !key || !data) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_set_weight: null argument");
    if (!h->raw_arena) {
        HIP_TRY(hipMalloc((void**)&h->raw_arena, h->raw_floats * sizeof(float)));
        HIP_TRY(hipMemsetAsync(h->raw_arena, 0, h->raw_floats * sizeof(float), (hipStream_t)stream));
    }
    if (!h->der_arena) {
        HIP_TRY(hipMalloc((void**)&h->der_arena, h->der_bytes));
        HIP_TRY(hipMemsetAsync(h->der_arena, 0, h->der_bytes, (hipStream_t)stream));
    }
    for (auto& t : h->raw) {
        if (t.key == key) {
            if (t.numel != numel)
                return fail(RNAMPNN_ERR_WEIGHTS, "weight '%s': expected %lld elements, got %lld", key,
                            (long long)t.numel, (long long)numel);
            HIP_TRY(hipMemcpyAsync(h->raw_arena + t.off, data, (size_t)numel * sizeof(float),
                                   is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, (hipStream_t)stream));
            t.set = true;
            h->finalized = false;
            return RNAMPNN_OK;
        }
    }
    return fail(RNAMPNN_ERR_WEIGHTS, "unknown state_dict key '%s'", key);
}

// The caller's flat f32 buffer of rnampnn_grad_numel() elements becomes the weight storage itself (tensor i at
// rnampnn_weight_offset(i)): an optimiser that updates it in place needs no upload, only rnampnn_finalize_weights
// before the next INFERENCE call (the bf16-mixed training kernels read nn.Linear.weight as stored).
extern "C" int rnampnn_use_weight_arena(rnampnn_handle h, float* arena, void* stream) {
    if (!h || !arena) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_use_weight_arena: null argument");
    if (((uintptr_t)arena & 15) != 0) return fail(RNAMPNN_ERR_BAD_ARG, "weight arena must be 16-byte aligned");
    if (h->raw_arena && !h->raw_external) (void)hipFree(h->raw_arena);
    h->raw_arena = arena;
    h->raw_external = true;
    t_wimg_clear(h->wimg);
    if (!h->der_arena) {
        HIP_TRY(hipMalloc((void**)&h->der_arena, h->der_bytes));
        HIP_TRY(hipMemsetAsync(h->der_arena, 0, h->der_bytes, (hipStream_t)stream));
    }
    for (auto& t : h->raw) t.set = true;
    h->finalized = false;
    return RNAMPNN_OK;
}

// ------------------------------------------------------------------------------------------
static inline float* rawp(rnampnn_ctx* c, int idx) { return c->raw_arena + c->raw[idx].off; }
template <typename T> static inline T* derp(rnampnn_ctx* c, size_t off) { return reinterpret_cast<T*>(c->der_arena + off); }

static void finalize_lin(rnampnn_ctx* c, const Lin& l, hipStream_t s) {
    launch_transpose(rawp(c, l.w), l.in, l.out, l.in, derp<float>(c, l.wt), l.out, s);
    if (c->cfg.precision == RNAMPNN_PREC_BF16)
        launch_convert_rows_bf16(rawp(c, l.w), l.in, l.out, l.in, l.in_pad, derp<bf16_t>(c, l.wb), s);
}
static void finalize_mlp2(rnampnn_ctx* c, const Mlp2& m, bool is_edge, hipStream_t s) {
    const float* w0 = rawp(c, m.w[0]);                 // [128][384] = [out][h_i | h_j | e]
    float* pq = derp<float>(c, m.pq_t);
    launch_transpose(w0, 3 * RN_D, RN_D, RN_D, pq, 256, s);                    // P part
    launch_transpose(w0 + RN_D, 3 * RN_D, RN_D, RN_D, pq + RN_D, 256, s);      // Q part
    launch_copy_bytes(derp<float>(c, m.pq_b), rawp(c, m.b[0]), RN_D * sizeof(float), s);       // (kernel copy: a launch error surfaces in hipGetLastError below)
    launch_transpose(w0 + 2 * RN_D, 3 * RN_D, RN_D, RN_D, derp<float>(c, m.wc_t), RN_D, s);
    if (m.depth > 1) launch_transpose(rawp(c, m.w[1]), RN_D, RN_D, RN_D, derp<float>(c, m.w2_t), RN_D, s);
    if (c->cfg.precision == RNAMPNN_PREC_BF16) {
        // node GEMM weights [P rows | Q rows] = W0[:, 0:128] and W0[:, 128:256], bf16 [256][128]
        const bool one = m.depth == 1;           // (edge MLP only: rnampnn_create admits no other depth-1 MLP on this path)
        launch_build_pq_image(w0, rawp(c, m.b[0]), one ? 1 : 0, derp<bf16_t>(c, m.pq_img), derp<float>(c, m.pq_bp), s);
        launch_build_mlp_image(w0 + 2 * RN_D, 3 * RN_D, one ? nullptr : rawp(c, m.w[1]), RN_D, one ? nullptr : rawp(c, m.b[1]),
                               is_edge ? 1 : 0, derp<bf16_t>(c, m.img), derp<float>(c, m.b2p), s);
    }
}

static void finalize_chain(rnampnn_ctx* c, const Chain& ch, const std::vector<Lin>& layers, hipStream_t s) {
    if (!ch.ok) return;
    bf16_t* dst = derp<bf16_t>(c, ch.img);
    for (size_t i = 0; i < layers.size(); ++i) {
        const Lin& l = layers[i];
        bool last = i + 1 == layers.size();
        int K = i == 0 ? ch.K0 : ch.H, N = last ? ch.NOUT : ch.H;
        launch_build_chain_image(rawp(c, l.w), l.in, K, N, l.out, i == 0 ? 0 : (last ? 2 : 1), dst, s);
        dst += (size_t)(N / 32) * (K / 16) * 512;
    }
    // derived arena was zeroed: bias beyond the real outputs stays 0
    launch_copy_bytes(derp<float>(c, ch.last_bias), rawp(c, layers.back().b), (size_t)(layers.back().out + 3) / 4 * 4 * sizeof(float), s);   // (padded to 16 B: the
                                                                                     // raw arena pads every tensor to 4 floats, NOUT >= 32)
}

extern "C" int rnampnn_finalize_weights(rnampnn_handle h, void* stream) {
    if (!h) return fail(RNAMPNN_ERR_BAD_ARG, "null handle");
    if (!h->raw_arena) return fail(RNAMPNN_ERR_WEIGHTS, "no weights have been set");
    for (auto& t : h->raw)
        if (!t.set) return fail(RNAMPNN_ERR_WEIGHTS, "weight '%s' has not been set", t.key.c_str());
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(h->der_arena, 0, h->der_bytes, s));
    finalize_lin(h, h->raw_project, s);
    for (Bert* b : {&h->emb, &h->post}) {
        for (auto& a : b->attn) {
            finalize_lin(h, a.qkv, s); finalize_lin(h, a.out, s);
            if (h->cfg.precision == RNAMPNN_PREC_BF16)
                launch_build_attn_images(rawp(h, a.qkv.w), rawp(h, a.out.w), derp<bf16_t>(h, a.img_qkv), derp<bf16_t>(h, a.img_out), s);
        }
        for (auto& l : b->ffn) finalize_lin(h, l, s);
    }
    for (auto& l : h->edge_embed) finalize_lin(h, l, s);
    for (auto& m : h->mpnn) { finalize_mlp2(h, m.msg, false, s); finalize_mlp2(h, m.edge, true, s); }
    for (auto& l : h->raw_ffn) finalize_lin(h, l, s);
    for (auto& l : h->readout) finalize_lin(h, l, s);
    finalize_chain(h, h->emb.chain, h->emb.ffn, s);
    finalize_chain(h, h->post.chain, h->post.ffn, s);
    finalize_chain(h, h->raw_chain, h->raw_ffn, s);
    finalize_chain(h, h->readout_chain, h->readout, s);
    if (h->cfg.precision == RNAMPNN_PREC_BF16)
        launch_build_embed_image(rawp(h, h->edge_embed[0].w), rawp(h, h->edge_embed[1].w), rawp(h, h->edge_embed[1].b),
                                 derp<bf16_t>(h, h->edge_embed_img), derp<float>(h, h->edge_embed_b1p), s);
    HIP_TRY(hipGetLastError());
    h->finalized = true;
    return RNAMPNN_OK;
}

// ------------------------------------------------------------------------------------------
struct Ws {                       // workspace carve (all offsets 256-byte aligned)
    int *len, *cu, *node_b, *nbr;
    float *geom, *geomh, *raw_p, *hA, *hB, *pq_e, *pq_m, *s0, *s1, *n0, *n1, *n2, *logits_p;
    float* coef;                  // fast path: per-RNA GraphNorm affine coefficients [B][256]
    bf16_t *p_e, *p_m, *q_e, *q_m;   // fast path: the f16 P / Q tables [(Nmax+1)][128] of the two first Linears (pq_* are the f32 path's [P | Q])
    void* e;                      // f32 or bf16 [Nmax*k][128]
    float* big;                   // [Nmax*k][128] f32 scratch for the stage API / edge taps
    size_t total;
};

static size_t carve(const rnampnn_ctx* c, int B, size_t Nmax, char* base, Ws* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return base ? base + o : (char*)nullptr; };
    size_t k = c->cfg.num_res_neighbours, F = c->fmax;
    size_t esz = c->cfg.precision == RNAMPNN_PREC_BF16 ? sizeof(bf16_t) : sizeof(float);
    Ws tmp;
    Ws& r = w ? *w : tmp;
    r.len = (int*)take(B * sizeof(int));
    r.cu = (int*)take((B + 1) * sizeof(int));
    r.node_b = (int*)take(Nmax * sizeof(int));
    r.nbr = (int*)take(Nmax * k * sizeof(int));
    r.geom = (float*)take((Nmax + B) * RN_GEOM * sizeof(float));
    r.geomh = (float*)take(c->cfg.precision == RNAMPNN_PREC_BF16 ? (Nmax + B) * RN_GEOMH * sizeof(float) : 0);
    r.raw_p = (float*)take(Nmax * RN_RAWP * sizeof(float));
    r.hA = (float*)take((Nmax + 1) * RN_D * sizeof(float));
    r.hB = (float*)take((Nmax + 1) * RN_D * sizeof(float));
    const bool fastp = c->cfg.precision == RNAMPNN_PREC_BF16;
    r.pq_e = (float*)take(fastp ? (Nmax + 1) * 2 * RN_D * sizeof(float) : (Nmax + 1) * 256 * sizeof(float));   // (fast path: the read-out stage API packs its input here)
    r.pq_m = (float*)take(fastp ? 0 : (Nmax + 1) * 256 * sizeof(float));
    r.coef = (float*)take((size_t)B * 256 * sizeof(float));
    r.p_e = (bf16_t*)take(fastp ? (Nmax + 1) * RN_D * sizeof(bf16_t) : 0);
    r.p_m = (bf16_t*)take(fastp ? (Nmax + 1) * RN_D * sizeof(bf16_t) : 0);
    r.q_e = (bf16_t*)take(fastp ? (Nmax + 1) * RN_D * sizeof(bf16_t) : 0);
    r.q_m = (bf16_t*)take(fastp ? (Nmax + 1) * RN_D * sizeof(bf16_t) : 0);
    r.s0 = (float*)take(Nmax * F * sizeof(float));
    r.s1 = (float*)take(Nmax * F * sizeof(float));
    r.n0 = (float*)take(Nmax * RN_D * sizeof(float));
    r.n1 = (float*)take(Nmax * RN_D * sizeof(float));
    r.n2 = (float*)take(Nmax * RN_D * sizeof(float));
    r.logits_p = (float*)take(Nmax * 4 * sizeof(float));
    r.e = (void*)take(c->cfg.precision == RNAMPNN_PREC_BF16 ? efrag_bytes((int)Nmax, (int)k) : Nmax * k * RN_D * esz);
    r.big = (float*)take(Nmax * k * RN_D * sizeof(float));
    r.total = off;
    return off;
}

extern "C" size_t rnampnn_workspace_bytes(rnampnn_handle h, int32_t B, int32_t T) {
    if (!h || B <= 0 || T <= 0) return 0;
    return carve(h, B, (size_t)B * T, nullptr, nullptr);
}
extern "C" size_t rnampnn_workspace_bytes_packed(rnampnn_handle h, int32_t B, int32_t N_total) {
    if (!h || B <= 0 || N_total <= 0) return 0;
    return carve(h, B, (size_t)N_total, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------
struct Run {                      // one call's launch context
    rnampnn_ctx* c;
    PackInfo pk;
    Ws w;
    hipStream_t s;
    bool fast;
    const int* ntot() const { return pk.cu + pk.B; }
};

static void gemm(Run& r, const Lin& l, const float* X, int ldx, float* Y, int ldy, const float* res = nullptr,
                 int ldres = 0, const float* X2 = nullptr, int ldx2 = 0, int K1 = -1) {
    rnampnn_ctx* c = r.c;
    int k1 = K1 < 0 ? l.in_pad : K1;
    int k2 = K1 < 0 ? 0 : l.in_pad - K1;
    if (r.fast && l.out >= 64)
        launch_gemm_bf16(r.ntot(), r.pk.Nmax, X, ldx, k1, X2, ldx2, k2, derp<bf16_t>(c, l.wb), rawp(c, l.b), l.out,
                         l.gelu ? 1 : 0, res, ldres, Y, ldy, r.s);
    else
        launch_gemm_f32(r.ntot(), r.pk.Nmax, X, ldx, k1, X2, ldx2, k2, derp<float>(c, l.wt), rawp(c, l.b), l.out,
                        l.gelu ? 1 : 0, res, ldres, Y, ldy, r.s);
}

// fused FFN chain (fast path); returns false when the shape has no fused kernel
static bool run_chain(Run& r, const Chain& ch, const std::vector<Lin>& layers, const float* X, int ldx, const float* X2,
                      int ldx2, float* Y, int ldy) {
    if (!r.fast || !ch.ok) return false;
    rnampnn_ctx* c = r.c;
    const float* bias[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (size_t i = 0; i + 1 < layers.size(); ++i) bias[i] = rawp(c, layers[i].b);
    bias[layers.size() - 1] = derp<float>(c, ch.last_bias);
    return launch_ffn_chain(r.ntot(), r.pk.Nmax, X, ldx, X2, ldx2, ch.K0, ch.H, ch.NH, ch.NOUT, derp<bf16_t>(c, ch.img),
                            bias, Y, ldy, ch.n_valid, r.s) == 0;
}

// RNABert.forward on packed rows: x [N][128] -> out [N][128] (x is clobbered when attention layers exist)
static int run_bert(Run& r, const Bert& b, float* x, float* out) {
    rnampnn_ctx* c = r.c;
    for (auto& a : b.attn) {
        if (r.fast && launch_attn_layer_rna(r.pk, x, derp<bf16_t>(c, a.img_qkv), rawp(c, a.qkv.b), derp<bf16_t>(c, a.img_out), rawp(c, a.out.b),
                                            b.heads, rawp(c, a.gn_scale), rawp(c, a.gn_shift), c->cfg.padding_len, r.s) == 0)
            continue;
        gemm(r, a.qkv, x, RN_D, r.w.s0, 3 * RN_D);
        if (!(r.fast && launch_attention_bf16(r.pk, r.w.s0, b.heads, r.w.n2, r.s) == 0) &&
            launch_attention_f32(r.pk, r.w.s0, b.heads, r.w.n2, r.s))
            return fail(RNAMPNN_ERR_UNSUPPORTED, "head dim unsupported");
        gemm(r, a.out, r.w.n2, RN_D, r.w.s1, RN_D, x, RN_D);                          // x + out_proj(attn)
        launch_graph_norm_packed(r.pk, r.w.s1, nullptr, x, rawp(c, a.gn_scale), rawp(c, a.gn_shift), c->cfg.padding_len, r.s);
    }
    if (run_chain(r, b.chain, b.ffn, x, RN_D, nullptr, 0, out, RN_D)) return RNAMPNN_OK;
    const float* cur = x;
    int ld = RN_D;
    float* bufs[2] = {r.w.s0, r.w.s1};
    for (size_t i = 0; i < b.ffn.size(); ++i) {
        const Lin& l = b.ffn[i];
        bool last = i + 1 == b.ffn.size();
        float* dst = last ? out : bufs[i & 1];
        gemm(r, l, cur, ld, dst, l.out);
        cur = dst; ld = l.out;
    }
    return RNAMPNN_OK;
}

static void run_ffn(Run& r, const Chain& ch, const std::vector<Lin>& ffn, const float* x, int ldx, float* out) {
    if (run_chain(r, ch, ffn, x, ldx, nullptr, 0, out, ffn.back().out)) return;
    const float* cur = x;
    int ld = ldx;
    float* bufs[2] = {r.w.s0, r.w.s1};
    for (size_t i = 0; i < ffn.size(); ++i) {
        bool last = i + 1 == ffn.size();
        float* dst = last ? out : bufs[i & 1];
        gemm(r, ffn[i], cur, ld, dst, ffn[i].out);
        cur = dst; ld = ffn[i].out;
    }
}

static NodeJob node_job(rnampnn_ctx* c, const Mlp2& m, bf16_t* p, bf16_t* q) {
    NodeJob j;
    j.img = derp<bf16_t>(c, m.pq_img); j.bias = derp<float>(c, m.pq_bp); j.p = p; j.q = q; j.p_efrag = m.depth == 1;
    return j;
}
// the per-residue parts [P | Q] of one MLP's first Linear: `edge` selects the workspace tables of the edge-update / message MLP
static void node_pq(Run& r, const Mlp2& m, const float* h, bool edge) {
    rnampnn_ctx* c = r.c;
    if (r.fast)     // the fused node kernel without residual / norm
        launch_node_update(r.pk, h, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 1,
                           node_job(c, m, edge ? r.w.p_e : r.w.p_m, edge ? r.w.q_e : r.w.q_m), NodeJob{}, r.s);
    else
        launch_gemm_f32(r.ntot(), r.pk.Nmax, h, RN_D, RN_D, nullptr, 0, 0, derp<float>(c, m.pq_t),
                        derp<float>(c, m.pq_b), 256, 0, nullptr, 0, edge ? r.w.pq_e : r.w.pq_m, 256, r.s);
}

static bool embed_fused_env() { const char* v = getenv("RNAMPNN_EMBED_FUSED"); return !(v && v[0] == '0'); }      // RNAMPNN_EMBED_FUSED=0: two launches (A/B; read per call)

static MpnnW32 w32(rnampnn_ctx* c, const Mlp2& m) {
    MpnnW32 w;
    w.wc_t = derp<float>(c, m.wc_t);
    w.w2_t = m.depth > 1 ? derp<float>(c, m.w2_t) : nullptr;
    w.b2 = m.depth > 1 ? rawp(c, m.b[1]) : nullptr;
    w.depth = m.depth;
    return w;
}
static MpnnWB wbf(rnampnn_ctx* c, const Mlp2& m) {
    MpnnWB w;
    w.img = derp<bf16_t>(c, m.img);
    w.b2p = derp<float>(c, m.b2p);
    return w;
}

// one fused step: [edge update with `we`] then [message + aggregation with `wm`]
static void mpnn_step(Run& r, const Mlp2* we, const Mlp2* wm, const float* h_in, float* h_pre, float* msg_out, bool embed_first = false) {
    rnampnn_ctx* c = r.c;
    int k = c->cfg.num_res_neighbours;
    bool timed = c->prof && c->ev_used + 2 <= c->ev.size() && (c->prof_seen++ % c->prof_stride) == 0;
    if (timed) { c->ev_kind[c->ev_used / 2] = (we && wm) ? 0 : 1; (void)hipEventRecord(c->ev[c->ev_used], r.s); }
    if (embed_first) {      // layer 1's message with the edge embedding computed in front of it (e0 written once, never re-read)
        launch_resmpnn_embed_bf16(r.pk, k, r.w.nbr, (bf16_t*)r.w.e, r.w.geomh, derp<bf16_t>(c, c->edge_embed_img), rawp(c, c->edge_embed[0].b),
                                  derp<float>(c, c->edge_embed_b1p), r.w.p_m, r.w.q_m, wbf(c, *wm), h_pre, h_in, r.s);
    } else if (r.fast) {
        launch_mpnn_bf16(r.pk, k, we != nullptr, wm != nullptr, r.w.nbr, (bf16_t*)r.w.e, r.w.p_e, r.w.q_e, r.w.p_m,
                         r.w.q_m, we ? wbf(c, *we) : MpnnWB{}, wm ? wbf(c, *wm) : MpnnWB{}, h_pre, msg_out, we && we->depth == 1,
                         wm ? h_in : nullptr, r.s);      // h_pre = h_in + mean of the messages (both paths)
    } else {
        MpnnW32 e32 = we ? w32(c, *we) : MpnnW32{}, m32 = wm ? w32(c, *wm) : MpnnW32{};
        launch_mpnn_f32(r.pk, k, we != nullptr, wm != nullptr, r.w.nbr, (float*)r.w.e, r.w.pq_e, r.w.pq_m, e32, m32,
                        h_in, h_pre, msg_out, r.s);
    }
    if (timed) { (void)hipEventRecord(c->ev[c->ev_used + 1], r.s); c->ev_used += 2; }
}

static void unpack_e(Run& r, float* dst) {
    int k = r.c->cfg.num_res_neighbours;
    if (r.fast) {
        launch_efrag_to_rows((const bf16_t*)r.w.e, r.ntot(), r.pk.Nmax, k, r.w.big, r.s);
        launch_unpack_edges(r.pk, k, r.w.big, r.w.nbr, dst, r.s);
    } else {
        launch_unpack_edges(r.pk, k, (const float*)r.w.e, r.w.nbr, dst, r.s);
    }
}

// mask != null: padded API tensors (B,T); cu_seqlens != null: packed input of n_total rows (T = longest RNA)
static int begin_run(Run& r, rnampnn_handle h, const float* mask, int B, int T, void* ws, size_t ws_bytes, void* stream,
                     const int32_t* cu_seqlens = nullptr, int n_total = 0) {
    if (!h) return fail(RNAMPNN_ERR_BAD_ARG, "null handle");
    if (!h->finalized) return fail(RNAMPNN_ERR_WEIGHTS, "weights not finalized (call rnampnn_finalize_weights)");
    if ((!mask && !cu_seqlens) || !ws || B <= 0 || T <= 0) return fail(RNAMPNN_ERR_BAD_ARG, "null pointer or non-positive B/T");
    const long long nmax = cu_seqlens ? (long long)n_total : (long long)B * T;
    if (nmax <= 0 || nmax > 0x3fffffffLL / (h->cfg.num_res_neighbours * 4))
        return fail(RNAMPNN_ERR_BAD_ARG, "row count out of range for 32-bit edge indexing; split the batch");
    size_t need = carve(h, B, (size_t)nmax, nullptr, nullptr);
    if (ws_bytes < need) return fail(RNAMPNN_ERR_WORKSPACE, "workspace %zu bytes < required %zu", ws_bytes, need);
    if (((uintptr_t)ws & 255) != 0) return fail(RNAMPNN_ERR_BAD_ARG, "workspace must be 256-byte aligned");
    r.c = h;
    r.s = (hipStream_t)stream;
    r.fast = h->cfg.precision == RNAMPNN_PREC_BF16;
    carve(h, B, (size_t)nmax, (char*)ws, &r.w);
    r.pk.len = r.w.len; r.pk.cu = r.w.cu; r.pk.node_b = r.w.node_b;
    r.pk.B = B; r.pk.T = T; r.pk.Nmax = (int)nmax; r.pk.packed_in = cu_seqlens ? 1 : 0;
    // the all-zero row Nmax of every gathered node table (phantom neighbour / invalid slot)
    size_t Nmax = r.pk.Nmax;
    ZeroRegions z{};
    auto add = [&](void* p, size_t bytes) { z.ptr[z.n] = p; z.words[z.n] = (unsigned)(bytes / 4); ++z.n; };
    add(r.w.hA + Nmax * RN_D, RN_D * sizeof(float));
    add(r.w.hB + Nmax * RN_D, RN_D * sizeof(float));
    if (r.fast) {   // fast-path tables are f16 [N+1][128]
        add(r.w.p_e + Nmax * RN_D, RN_D * sizeof(bf16_t));
        add(r.w.p_m + Nmax * RN_D, RN_D * sizeof(bf16_t));
        add(r.w.q_e + Nmax * RN_D, RN_D * sizeof(bf16_t));
        add(r.w.q_m + Nmax * RN_D, RN_D * sizeof(bf16_t));
    } else {
        add(r.w.pq_e + Nmax * 256, 256 * sizeof(float));
        add(r.w.pq_m + Nmax * 256, 256 * sizeof(float));
    }
    if (cu_seqlens) launch_lengths_from_cu(cu_seqlens, r.pk, r.s);
    else launch_lengths(mask, r.pk, r.s);
    launch_zero_regions(z, r.s);
    return RNAMPNN_OK;
}

// packed outputs of the packed-input entry point (null for the padded API)
struct PackedOut { float* logits; float* embedding; int n_total; };
static int forward_core(Run& r, rnampnn_handle h, const RnaMpnnForwardIO* io, const PackedOut* po);

extern "C" int rnampnn_forward(rnampnn_handle h, const RnaMpnnForwardIO* io, void* ws, size_t ws_bytes, void* stream) {
    if (!io || !io->coords) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_forward: null io/coords");
    if (io->stop_after == 0 && !io->logits && !io->embedding)
        return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_forward: logits (or embedding) output required");
    Run r;
    int rc = begin_run(r, h, io->mask, io->B, io->T, ws, ws_bytes, stream);
    if (rc) return rc;
    return forward_core(r, h, io, nullptr);
}

extern "C" int rnampnn_forward_packed(rnampnn_handle h, const float* coords_packed, const int32_t* cu_seqlens, int32_t B,
                                      int32_t N_total, int32_t T_max, int32_t T_norm, float* logits_packed,
                                      float* embedding_packed, void* ws, size_t ws_bytes, void* stream) {
    if (!coords_packed || !cu_seqlens || (!logits_packed && !embedding_packed))
        return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_forward_packed: null input or no output");
    Run r;
    int rc = begin_run(r, h, nullptr, B, T_max, ws, ws_bytes, stream, cu_seqlens, N_total);
    if (rc) return rc;
    RnaMpnnForwardIO io;
    memset(&io, 0, sizeof(io));
    io.coords = coords_packed; io.B = B; io.T = T_max; io.T_norm = T_norm;
    PackedOut po{logits_packed, embedding_packed, N_total};
    return forward_core(r, h, &io, &po);
}

static int forward_core(Run& r, rnampnn_handle h, const RnaMpnnForwardIO* io, const PackedOut* po) {
    int rc = RNAMPNN_OK;
    rnampnn_ctx* c = h;
    const RnaMpnnConfig& g = c->cfg;
    if (io->T > g.padding_len)
        return fail(RNAMPNN_ERR_T_GT_P, "max_len %d exceeds padding_len %d", io->T, g.padding_len);
    int k = g.num_res_neighbours, L = g.num_res_mpnn_layers;
    int t_norm = io->T_norm > 0 ? io->T_norm : io->T;
    if (t_norm < io->T) return fail(RNAMPNN_ERR_BAD_ARG, "T_norm %d < T %d", t_norm, io->T);
    Ws& w = r.w;
    hipStream_t s = r.s;

    // ---- ResFeature.forward (feature.py:588-592)
    launch_geom(io->coords, r.pk, io->raw, w.raw_p, w.geom, r.fast ? w.geomh : nullptr, s);
    const bool fused_first = r.fast && io->stop_after != 1;
    {   // raw_project (feature.py:183, 28 -> 128) in exact f32 on both paths: the GraphNorm behind the embedding stack removes the common part of
        // its output and amplifies the rounding of the inputs (bf16 operands here cost 4 % of h0: tools/tap_errors.py); 0.1 GFLOP at the C2 batch
        const Lin& l = c->raw_project;
        launch_gemm_f32(r.ntot(), r.pk.Nmax, w.raw_p, RN_RAWP, l.in_pad, nullptr, 0, 0, derp<float>(c, l.wt), rawp(c, l.b), l.out,
                        l.gelu ? 1 : 0, nullptr, 0, w.n0, RN_D, r.s);
    }
    rc = run_bert(r, c->emb, w.n0, w.n1);
    if (!rc) {
        if (fused_first)    // GraphNorm + the [P | Q] projection of layer 1's message MLP in one pass
            launch_node_update(r.pk, w.n1, nullptr, rawp(c, c->feat_gn_scale), rawp(c, c->feat_gn_shift), t_norm, w.coef, w.hA, 1,
                               node_job(c, c->mpnn[0].msg, w.p_m, w.q_m), NodeJob{}, r.s);
        else
            launch_graph_norm_packed(r.pk, w.n1, nullptr, w.hA, rawp(c, c->feat_gn_scale), rawp(c, c->feat_gn_shift), t_norm, r.s);
    }
    if (rc) return rc;
    if (launch_knn(io->coords, r.pk, k, w.nbr, io->edge_index, s))
        return fail(RNAMPNN_ERR_UNSUPPORTED, "max_len %d too long for the LDS-resident k-NN row", io->T);
    const bool embed_first = fused_first && L >= 1 && !io->e0 && resmpnn_covers(k, false, false) && embed_fused_env();
    if (embed_first) { }
    else if (r.fast)
        launch_edge_embed_bf16(r.pk, k, w.geomh, w.nbr, derp<bf16_t>(c, c->edge_embed_img), rawp(c, c->edge_embed[0].b),
                               derp<float>(c, c->edge_embed_b1p), (bf16_t*)w.e, s);
    else
        launch_edge_embed_f32(r.pk, k, w.geom, w.nbr, derp<float>(c, c->edge_embed[0].wt), rawp(c, c->edge_embed[0].b),
                              g.depth_res_edge_feature > 1 ? derp<float>(c, c->edge_embed[1].wt) : nullptr,
                              g.depth_res_edge_feature > 1 ? rawp(c, c->edge_embed[1].b) : nullptr,
                              g.depth_res_edge_feature, (float*)w.e, s);
    if (io->h0) launch_unpack_nodes(r.pk, w.hA, RN_D, RN_D, io->h0, s);
    if (io->e0) unpack_e(r, io->e0);
    if (io->stop_after == 1) { HIP_TRY(hipGetLastError()); return RNAMPNN_OK; }

    // ---- L x ResMPNN.forward (mpnn.py:283-294), edge update of layer l fused with the message of l+1
    if (!fused_first) node_pq(r, c->mpnn[0].msg, w.hA, false);
    bool edge_pending = false;                                 // layer l-1's edge update not yet applied
    for (int l = 0; l < L; ++l) {
        mpnn_step(r, edge_pending ? &c->mpnn[l - 1].edge : nullptr, &c->mpnn[l].msg, w.hA, w.hB, nullptr, l == 0 && embed_first);
        bool tap_e = io->tap_layer == l + 1 && io->e_layer;
        edge_pending = l + 1 < L;                              // layer L's edge update is dead work
        const bool need_e = edge_pending || tap_e, need_m = l + 1 < L;
        const float* gsc = rawp(c, c->mpnn[l].gn_scale);
        const float* gsh = rawp(c, c->mpnn[l].gn_shift);
        if (r.fast && (need_e || need_m)) {
            // the bf16 kernel wrote h + agg; GraphNorm + both [P | Q] projections in one kernel
            const NodeJob je = need_e ? node_job(c, c->mpnn[l].edge, w.p_e, w.q_e) : NodeJob{};
            const NodeJob jm = need_m ? node_job(c, c->mpnn[l + 1].msg, w.p_m, w.q_m) : NodeJob{};
            launch_node_update(r.pk, w.hB, nullptr, gsc, gsh, t_norm, w.coef, w.hA, (need_e && need_m) ? 2 : 1, need_e ? je : jm, jm, s);
        } else {
            // (both kernel families write h + agg)
            launch_graph_norm_packed(r.pk, w.hB, nullptr, w.hA, gsc, gsh, t_norm, s);
            if (need_e) node_pq(r, c->mpnn[l].edge, w.hA, true);
            if (need_m) node_pq(r, c->mpnn[l + 1].msg, w.hA, false);
        }
        if (io->tap_layer == l + 1) {
            if (io->h_layer) launch_unpack_nodes(r.pk, w.hA, RN_D, RN_D, io->h_layer, s);
            if (tap_e) {
                mpnn_step(r, &c->mpnn[l].edge, nullptr, w.hA, w.hB, nullptr);
                edge_pending = false;
                unpack_e(r, io->e_layer);
            }
        }
    }
    // ---- post fusion, raw embedding, readout (rnampnn.py:179-181)
    float* raw_emb = w.n2;
    rc = run_bert(r, c->post, w.hA, w.n0);                     // h_post -> n0
    if (rc) return rc;
    run_ffn(r, c->raw_chain, c->raw_ffn, w.raw_p, RN_RAWP, w.n1);
    launch_graph_norm_packed(r.pk, w.n1, nullptr, w.n2, rawp(c, c->rawffn_gn_scale), rawp(c, c->rawffn_gn_shift), t_norm, s);   // raw_emb -> n2
    if (io->h_post) launch_unpack_nodes(r.pk, w.n0, RN_D, RN_D, io->h_post, s);
    if (io->raw_emb) launch_unpack_nodes(r.pk, raw_emb, RN_D, RN_D, io->raw_emb, s);
    if (io->embedding) {
        launch_unpack_nodes_strided(r.pk, w.n0, RN_D, RN_D, io->embedding, 2 * RN_D, 0, s);
        launch_unpack_nodes_strided(r.pk, raw_emb, RN_D, RN_D, io->embedding, 2 * RN_D, RN_D, s);
    }
    if (po && po->embedding) {      // packed rows: cat(h_post, raw_emb) by two strided copies
        HIP_TRY(hipMemcpy2DAsync(po->embedding, 2 * RN_D * sizeof(float), w.n0, RN_D * sizeof(float), RN_D * sizeof(float),
                                 po->n_total, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpy2DAsync(po->embedding + RN_D, 2 * RN_D * sizeof(float), raw_emb, RN_D * sizeof(float),
                                 RN_D * sizeof(float), po->n_total, hipMemcpyDeviceToDevice, s));
    }
    const bool want_logits = io->logits || (po && po->logits);
    float* logits_rows = (po && po->logits) ? po->logits : w.logits_p;      // packed output is written in place
    if (want_logits && run_chain(r, c->readout_chain, c->readout, w.n0, RN_D, raw_emb, RN_D, logits_rows, 4)) {
        if (io->logits) launch_unpack_nodes(r.pk, logits_rows, 4, 4, io->logits, s);
    } else if (want_logits) {
        const float* cur = nullptr;
        int ld = 0;
        float* bufs[2] = {w.s0, w.s1};
        for (size_t i = 0; i < c->readout.size(); ++i) {
            const Lin& l = c->readout[i];
            bool last = i + 1 == c->readout.size();
            float* dst = last ? logits_rows : bufs[i & 1];
            if (i == 0) gemm(r, l, w.n0, RN_D, dst, l.out, nullptr, 0, raw_emb, RN_D, RN_D);   // cat(h_post, raw_emb)
            else gemm(r, l, cur, ld, dst, l.out);
            cur = dst; ld = l.out;
        }
        if (io->logits) launch_unpack_nodes(r.pk, logits_rows, 4, 4, io->logits, s);
    }
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int rnampnn_mpnn_layer(rnampnn_handle h, int32_t layer, const float* h_in, const float* e_in,
                                  const int64_t* edge_index, const float* mask, int32_t B, int32_t T, int32_t T_norm,
                                  float* msg_out, float* h_out, float* e_out, void* ws, size_t ws_bytes, void* stream) {
    if (!h_in || !e_in || !edge_index) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_mpnn_layer: null input");
    Run r;
    int rc = begin_run(r, h, mask, B, T, ws, ws_bytes, stream);
    if (rc) return rc;
    rnampnn_ctx* c = h;
    if (layer < 0 || layer >= c->cfg.num_res_mpnn_layers) return fail(RNAMPNN_ERR_BAD_ARG, "layer %d out of range", layer);
    int k = c->cfg.num_res_neighbours;
    int t_norm = T_norm > 0 ? T_norm : T;
    Ws& w = r.w;
    hipStream_t s = r.s;
    launch_pack_nodes(r.pk, h_in, RN_D, w.hA, RN_D, s);
    launch_pack_index(r.pk, k, edge_index, w.nbr, s);
    if (r.fast) {
        launch_pack_edges(r.pk, k, e_in, w.big, s);
        launch_rows_to_efrag(w.big, r.ntot(), r.pk.Nmax, k, w.nbr, (bf16_t*)w.e, s);
    } else {
        launch_pack_edges(r.pk, k, e_in, (float*)w.e, s);
    }
    const MpnnLayer& m = c->mpnn[layer];
    node_pq(r, m.msg, w.hA, false);
    float* msg_p = msg_out ? w.big : nullptr;
    mpnn_step(r, nullptr, &m.msg, w.hA, w.hB, msg_p);
    if (msg_out) launch_unpack_edges(r.pk, k, w.big, nullptr, msg_out, s);
    if (h_out || e_out) {
        launch_graph_norm_packed(r.pk, w.hB, nullptr, w.hA, rawp(c, m.gn_scale), rawp(c, m.gn_shift), t_norm, s);
        if (h_out) launch_unpack_nodes(r.pk, w.hA, RN_D, RN_D, h_out, s);
        if (e_out) {
            node_pq(r, m.edge, w.hA, true);
            mpnn_step(r, &m.edge, nullptr, w.hA, w.hB, nullptr);
            unpack_e(r, e_out);
        }
    }
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

extern "C" int rnampnn_graph_norm(const float* x, const float* mask, const float* scale, const float* shift,
                                  int32_t B, int32_t T, int32_t T_tot, int32_t D, float* y, void* stream) {
    if (!x || !mask || !scale || !shift || !y || B <= 0 || T <= 0 || D <= 0)
        return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_graph_norm: bad argument");
    if (T_tot <= 0) T_tot = T;
    if (T_tot < T) return fail(RNAMPNN_ERR_BAD_ARG, "T_tot %d < T %d", T_tot, T);
    launch_graph_norm_padded(x, mask, scale, shift, B, T, T_tot, D, y, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

extern "C" int rnampnn_rnabert(rnampnn_handle h, int32_t which, const float* x, const float* mask, int32_t B, int32_t T,
                               float* y, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_rnabert: null tensor");
    Run r;
    int rc = begin_run(r, h, mask, B, T, ws, ws_bytes, stream);
    if (rc) return rc;
    if (T > h->cfg.padding_len) return fail(RNAMPNN_ERR_T_GT_P, "max_len %d exceeds padding_len %d", T, h->cfg.padding_len);
    launch_pack_nodes(r.pk, x, RN_D, r.w.hA, RN_D, r.s);
    rc = run_bert(r, which == 0 ? h->emb : h->post, r.w.hA, r.w.n0);
    if (rc) return rc;
    launch_unpack_nodes(r.pk, r.w.n0, RN_D, RN_D, y, r.s);
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

extern "C" int rnampnn_raw_ffn(rnampnn_handle h, const float* raw, const float* mask, int32_t B, int32_t T, int32_t T_norm,
                               float* y, void* ws, size_t ws_bytes, void* stream) {
    if (!raw || !y) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_raw_ffn: null tensor");
    Run r;
    int rc = begin_run(r, h, mask, B, T, ws, ws_bytes, stream);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(r.w.raw_p, 0, (size_t)r.pk.Nmax * RN_RAWP * sizeof(float), r.s));
    launch_pack_nodes(r.pk, raw, RN_RAW, r.w.raw_p, RN_RAWP, r.s);
    run_ffn(r, h->raw_chain, h->raw_ffn, r.w.raw_p, RN_RAWP, r.w.n1);
    launch_graph_norm_packed(r.pk, r.w.n1, nullptr, r.w.n2, rawp(h, h->rawffn_gn_scale), rawp(h, h->rawffn_gn_shift),
                             T_norm > 0 ? T_norm : T, r.s);
    launch_unpack_nodes(r.pk, r.w.n2, RN_D, RN_D, y, r.s);
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

extern "C" int rnampnn_readout(rnampnn_handle h, const float* emb, const float* mask, int32_t B, int32_t T,
                               float* logits, void* ws, size_t ws_bytes, void* stream) {
    if (!emb || !logits) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_readout: null tensor");
    Run r;
    int rc = begin_run(r, h, mask, B, T, ws, ws_bytes, stream);
    if (rc) return rc;
    Ws& w = r.w;
    launch_pack_nodes(r.pk, emb, 2 * RN_D, w.pq_e, 2 * RN_D, r.s);
    const float* cur = w.pq_e;
    int ld = 2 * RN_D;
    float* bufs[2] = {w.s0, w.s1};
    for (size_t i = 0; i < h->readout.size(); ++i) {
        const Lin& l = h->readout[i];
        bool last = i + 1 == h->readout.size();
        float* dst = last ? w.logits_p : bufs[i & 1];
        gemm(r, l, cur, ld, dst, l.out);
        cur = dst; ld = l.out;
    }
    launch_unpack_nodes(r.pk, w.logits_p, 4, 4, logits, r.s);
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

extern "C" int rnampnn_argmax_recovery(const float* logits, const float* mask, const int32_t* labels, int32_t B, int32_t T,
                                       int8_t* pred, int32_t* correct, int32_t* valid, void* stream) {
    if (!logits || !mask || B <= 0 || T <= 0) return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_argmax_recovery: bad argument");
    launch_argmax_recovery(logits, mask, labels, B, T, pred, correct, valid, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

extern "C" int rnampnn_sample(const float* logits, const float* mask, int32_t B, int32_t T, float temperature,
                              int32_t n_samples, uint64_t seed, int8_t* out, void* stream) {
    if (!logits || !mask || !out || B <= 0 || T <= 0 || n_samples <= 0)
        return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_sample: bad argument");
    if (!(temperature > 0.f)) return fail(RNAMPNN_ERR_BAD_ARG, "temperature must be > 0");
    launch_sample(logits, mask, B, T, temperature, n_samples, seed, nullptr, out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

extern "C" int rnampnn_sample_dev_seed(const float* logits, const float* mask, int32_t B, int32_t T, float temperature,
                                       int32_t n_samples, const uint64_t* seed_device, int8_t* out, void* stream) {
    if (!logits || !mask || !out || !seed_device || B <= 0 || T <= 0 || n_samples <= 0)
        return fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_sample_dev_seed: bad argument");
    if (!(temperature > 0.f)) return fail(RNAMPNN_ERR_BAD_ARG, "temperature must be > 0");
    launch_sample(logits, mask, B, T, temperature, n_samples, 0, seed_device, out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return RNAMPNN_OK;
}

#include "train.inc"
