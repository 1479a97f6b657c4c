// Gradient-boosted-tree read-out on the device: the reference classifies the 256-d residue embeddings with
// `xgb.XGBClassifier(objective='multi:softmax', num_class=4, n_estimators=150, max_depth=8)` (rnampnn/model/rnampnn.py:136-145), predict =
// `self.xgb_readout.predict(embedding)` (rnampnn.py:297-298).  XGBoost itself is a third-party dependency (requirements.txt: xgboost~=2.1.1),
// not installed here, and the reference ships no fitted model: PARITY UNPINNED.  What is restated is XGBoost's published prediction rule
// for `gbtree` models: per tree, start at node 0; at an internal node go LEFT iff x[split_index] < split_condition, a missing value (NaN)
// follows `default_left`; a leaf contributes its value to the margin of the tree's class (`tree_info`); multi:softmax predicts the argmax
// of the margins (first maximum).  The trees arrive as the arrays of XGBoost's JSON model format (`Booster.save_model("m.json")`).
#include "../../include/rnampnn_hip.h"
#include "rnampnn_internal.h"

#include <cstdio>
#include <cstring>
#include <vector>

struct rnampnn_gbdt {
    int num_trees = 0, num_class = 0, num_feature = 0, total_nodes = 0;
    float base_score = 0.f;
    int *d_off = nullptr, *d_cls = nullptr;
    int4* d_nodes = nullptr;           // {left, right, feature | default_left << 31, threshold / leaf value bits}
};

static thread_local char gb_err[256] = "";
extern "C" const char* rnampnn_gbdt_last_error(void) { return gb_err; }
static int gb_fail(int code, const char* msg) { snprintf(gb_err, sizeof(gb_err), "%s", msg); return code; }

// One WAVE per row, one LANE per tree (64 trees at a time): the row's features sit in LDS (every lane reads them at data-dependent
// offsets), a node is ONE 16-byte record {left, right, feature | default_left << 31, threshold bits} (a walk is a chain of dependent
// L2 reads - one per level instead of five), and the leaves of a 64-tree chunk go through LDS to the class lanes, which add them IN
// TREE ORDER (the association order of a margin is the model's tree order: bit-reproducible, and the CPU restatement's).
__global__ void __launch_bounds__(256) k_gbdt_predict(const float* __restrict__ X, int n_rows, int ldx, int num_feature, int num_class, int num_trees,
        const int* __restrict__ off, const int* __restrict__ cls, const int4* __restrict__ nodes, float base_score,
        float* __restrict__ margin, int* __restrict__ argmax_out) {
    extern __shared__ float sm[];                      // [4 waves][num_feature + 64 leaves + 64 margins] | tree classes (bytes)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = num_feature + 128;
    float* xs = sm + wave * per_wave;
    float* leaf = xs + num_feature;
    float* ms = leaf + 64;
    unsigned char* tcls = reinterpret_cast<unsigned char*>(sm + 4 * per_wave);
    for (int t = threadIdx.x; t < num_trees; t += 256) tcls[t] = (unsigned char)cls[t];
    __syncthreads();
    for (int row = blockIdx.x * 4 + wave; row < n_rows; row += gridDim.x * 4) {
        for (int f = lane; f < num_feature; f += 64) xs[f] = X[(size_t)row * ldx + f];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float m = base_score;
        for (int c0 = 0; c0 < num_trees; c0 += 64) {
            const int t = c0 + lane;
            float v = 0.f;
            if (t < num_trees) {
                const int4* tn = nodes + off[t];
                int4 nd = tn[0];
                while (nd.x >= 0) {
                    const float x = xs[nd.z & 0x7fffffff];
                    const bool go_left = (x != x) ? (nd.z < 0) : (x < __int_as_float(nd.w));
                    nd = tn[go_left ? nd.x : nd.y];
                }
                v = __int_as_float(nd.w);              // leaf value (XGBoost stores it in split_conditions)
            }
            leaf[lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < num_class) {
                const int n = min(64, num_trees - c0);
                for (int u = 0; u < n; ++u)
                    if (tcls[c0 + u] == lane) m += leaf[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < num_class) {
            ms[lane] = m;
            if (margin) margin[(size_t)row * num_class + lane] = m;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && argmax_out) {
            int best = 0;
            float bv = ms[0];
            for (int k = 1; k < num_class; ++k) { const float q = ms[k]; if (q > bv) { bv = q; best = k; } }
            argmax_out[row] = best;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int rnampnn_gbdt_create(int32_t num_trees, int32_t num_class, int32_t num_feature, float base_score, const int32_t* tree_offsets,
                                   const int32_t* tree_class, const int32_t* left_children, const int32_t* right_children,
                                   const int32_t* split_indices, const float* split_conditions, const uint8_t* default_left,
                                   rnampnn_gbdt_handle* out) {
    if (!out || !tree_offsets || !tree_class || !left_children || !right_children || !split_indices || !split_conditions || !default_left)
        return gb_fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_gbdt_create: null argument");
    if (num_trees <= 0 || num_class <= 0 || num_class > 64 || num_feature <= 0 || num_class > 255) return gb_fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_gbdt_create: bad model dimensions");
    const int total = tree_offsets[num_trees];
    if (tree_offsets[0] != 0 || total <= 0) return gb_fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_gbdt_create: tree_offsets must start at 0");
    // host-side validation: every walk must terminate inside its tree and read existing features (the kernel trusts the arrays)
    for (int t = 0; t < num_trees; ++t) {
        const int b = tree_offsets[t], n = tree_offsets[t + 1] - b;
        if (n <= 0 || tree_class[t] < 0 || tree_class[t] >= num_class) return gb_fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_gbdt_create: empty tree or class id out of range");
        for (int i = 0; i < n; ++i) {
            const int l = left_children[b + i], r = right_children[b + i];
            if (l < 0) continue;                       // leaf
            // children come after their parent in XGBoost's node order: the walk strictly advances, hence terminates
            if (l <= i || r <= i || l >= n || r >= n) return gb_fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_gbdt_create: child index out of order / range");
            if (split_indices[b + i] < 0 || split_indices[b + i] >= num_feature) return gb_fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_gbdt_create: split index out of range");
        }
    }
    rnampnn_gbdt* g = new rnampnn_gbdt();
    g->num_trees = num_trees; g->num_class = num_class; g->num_feature = num_feature; g->total_nodes = total; g->base_score = base_score;
    std::vector<int4> packed((size_t)total);
    for (int i = 0; i < total; ++i) {
        int4 n;
        n.x = left_children[i]; n.y = right_children[i];
        n.z = (split_indices[i] & 0x7fffffff) | (default_left[i] ? (int)0x80000000u : 0);
        memcpy(&n.w, &split_conditions[i], sizeof(float));
        packed[i] = n;
    }
    auto up = [&](void** d, const void* h, size_t bytes) { return hipMalloc(d, bytes) == hipSuccess && hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess; };
    const bool ok = up((void**)&g->d_off, tree_offsets, sizeof(int) * (num_trees + 1)) && up((void**)&g->d_cls, tree_class, sizeof(int) * num_trees) &&
                    up((void**)&g->d_nodes, packed.data(), sizeof(int4) * total);
    if (!ok) { rnampnn_gbdt_destroy(g); return gb_fail(RNAMPNN_ERR_HIP, "rnampnn_gbdt_create: device allocation / upload failed"); }
    *out = g;
    return RNAMPNN_OK;
}
extern "C" int rnampnn_gbdt_destroy(rnampnn_gbdt_handle g) {
    if (!g) return RNAMPNN_OK;
    for (void* p : {(void*)g->d_off, (void*)g->d_cls, (void*)g->d_nodes})
        if (p) (void)hipFree(p);
    delete g;
    return RNAMPNN_OK;
}
extern "C" int rnampnn_gbdt_predict(rnampnn_gbdt_handle g, const float* X, int32_t n_rows, int32_t ldx, float* margin, int32_t* argmax_out,
                                    void* stream) {
    if (!g || !X || n_rows <= 0 || ldx < g->num_feature || (!margin && !argmax_out)) return gb_fail(RNAMPNN_ERR_BAD_ARG, "rnampnn_gbdt_predict: bad argument");
    const size_t lds = (size_t)4 * (g->num_feature + 128) * sizeof(float) + (size_t)(g->num_trees + 3) / 4 * 4;
    if (lds > 64 * 1024) return gb_fail(RNAMPNN_ERR_UNSUPPORTED, "rnampnn_gbdt_predict: model too wide for the LDS-resident row (num_feature / num_trees)");
    int grid = (n_rows + 3) / 4;
    const int cap = 16 * rn_num_cus();
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_gbdt_predict, dim3(grid), dim3(256), lds, (hipStream_t)stream, X, n_rows, ldx, g->num_feature, g->num_class,
                       g->num_trees, g->d_off, g->d_cls, g->d_nodes, g->base_score, margin, argmax_out);
    if (hipGetLastError() != hipSuccess) return gb_fail(RNAMPNN_ERR_HIP, "rnampnn_gbdt_predict: launch failed");
    return RNAMPNN_OK;
}
