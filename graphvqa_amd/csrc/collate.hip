// Scene-graph collate on the host, behind the C ABI: the topology and token rules of the reference's converter
// (/root/reference gqa_dataset_entry.py:231-332) and the block-diagonal batching of PyG's Batch.from_data_list (:654), over
// PRE-TOKENISED, flattened scene graphs (string -> id lookups are vocabulary work and stay with the caller), so that a loader
// feeds the path without a Python loop over nodes and edges.
//
// Input, per batch of B graphs (objects of a graph in the converter's node order = object ids sorted as strings):
//   graph_obj_ptr[B+1]          objects of graph g = [ptr[g], ptr[g+1])            (an empty graph: the converter's 2-node dummy, :196-224)
//   name_tok[O]                 object name token
//   attr_ptr[O+1], attr_tok[]   tokens of the object's DISTINCT attribute strings (the converter de-duplicates strings, :282 -- two different
//                               strings may share a token id, e.g. <unk>, and both are kept: so the tokeniser, which sees the strings, de-duplicates)
//   rel_ptr[O+1], rel_dst[], rel_tok[]   outgoing relations: destination object as a LOCAL index inside its graph, relation token
// Output (caller-owned host buffers sized by gvqa_scene_graph_collate_sizes):
//   x_tokens [N,12] (name, attributes, pad), edge_index [2,E] (row 0 source, row 1 destination, batched node ids), edge_tokens [E],
//   added_sym_edge [A] (batched edge ids of the reverse edges the converter adds), batch [N], graph_ptr / edge_ptr [B+1] (the
//   loader-side layout gvqa_graph_finalize_host takes), the largest in-degree.
// Per node, in order: one self-loop (`self_tok`), then for each relation the forward edge and -- only if the reverse pair is
// not among the graph's relation pairs -- a reverse edge with the same token, recorded in added_sym_edge (:318-332).
#include <algorithm>
#include <vector>

#include "common.h"

namespace gvqa {

constexpr int MAX_OBJ_TOKENS = 12;      // gqa_dataset_entry.py:268

static inline uint64_t pair_key(int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

// The (source, destination) pairs of one graph's relations, rebuilt per graph without allocating: a bit matrix for graphs of up
// to 64 objects (one 64-bit word per source), a sorted key vector beyond (a node-allocating hash set here was 30 of the 41 ms of
// a 2048-graph collate).
struct PairSet {
    uint64_t adj[64];
    std::vector<uint64_t> keys;
    bool small = true;
    void begin(int64_t n) {
        small = n <= 64;
        if (small) std::fill(adj, adj + n, 0ull);
        else keys.clear();
    }
    void insert(int a, int b) {
        if (small) adj[a] |= 1ull << b;
        else keys.push_back(pair_key(a, b));
    }
    void finish() {
        if (!small) std::sort(keys.begin(), keys.end());
    }
    bool count(int a, int b) const {
        if (small) return (adj[a] >> b) & 1ull;
        return std::binary_search(keys.begin(), keys.end(), pair_key(a, b));
    }
};

// nodes / edges / added reverse edges of graph g
static int graph_sizes(int64_t g, const int32_t* obj_ptr, const int32_t* rel_ptr, const int32_t* rel_dst, int64_t& n, int64_t& e, int64_t& a,
                       PairSet& pairs) {
    const int o0 = obj_ptr[g], o1 = obj_ptr[g + 1];
    GVQA_REQUIRE(o1 >= o0, GVQA_E_INVALID, "scene_graph_collate: graph_obj_ptr must be non-decreasing");
    n = o1 - o0;
    if (n == 0) { n = 2; e = 4; a = 0; return GVQA_OK; }         // the dummy graph: two nodes pointing at each other
    pairs.begin(n);
    for (int o = o0; o < o1; ++o)
        for (int r = rel_ptr[o]; r < rel_ptr[o + 1]; ++r) {
            GVQA_REQUIRE(rel_dst[r] >= 0 && rel_dst[r] < n, GVQA_E_GRAPH, "scene_graph_collate: relation %d of graph %lld points outside its graph", r, (long long)g);
            pairs.insert(o - o0, rel_dst[r]);
        }
    pairs.finish();
    e = n;
    a = 0;
    for (int o = o0; o < o1; ++o)
        for (int r = rel_ptr[o]; r < rel_ptr[o + 1]; ++r) {
            ++e;
            if (!pairs.count(rel_dst[r], o - o0)) { ++e; ++a; }
        }
    return GVQA_OK;
}

}  // namespace gvqa

extern "C" {

using namespace gvqa;

int gvqa_scene_graph_collate_sizes(int64_t B, const int32_t* graph_obj_ptr, const int32_t* rel_ptr, const int32_t* rel_dst,
                                   int64_t* sizes /* [3]: N, E, A */) {
    GVQA_REQUIRE(B >= 0 && graph_obj_ptr && sizes && (B == 0 || graph_obj_ptr[B] == 0 || (rel_ptr && (rel_ptr[graph_obj_ptr[B]] == 0 || rel_dst))),
                 GVQA_E_INVALID, "scene_graph_collate_sizes: null argument");
    int64_t N = 0, E = 0, A = 0;
    PairSet pairs;
    for (int64_t g = 0; g < B; ++g) {
        int64_t n, e, a;
        const int rc = graph_sizes(g, graph_obj_ptr, rel_ptr, rel_dst, n, e, a, pairs);
        if (rc) return rc;
        N += n; E += e; A += a;
    }
    GVQA_REQUIRE(N < (1ll << 31) && E < (1ll << 31), GVQA_E_INVALID, "scene_graph_collate_sizes: batch too large");
    sizes[0] = N; sizes[1] = E; sizes[2] = A;
    return GVQA_OK;
}

int gvqa_scene_graph_collate(int64_t B, const int32_t* graph_obj_ptr, const int64_t* name_tok, const int32_t* attr_ptr,
                             const int64_t* attr_tok, const int32_t* rel_ptr, const int32_t* rel_dst, const int64_t* rel_tok,
                             int64_t pad_tok, int64_t self_tok, int64_t unk_tok, int64_t N, int64_t E, int64_t A, int64_t* x_tokens,
                             int64_t* edge_index, int64_t* edge_tokens, int64_t* added_sym_edge, int64_t* batch, int32_t* graph_ptr,
                             int32_t* edge_ptr, int32_t* max_in_degree) {
    GVQA_REQUIRE(B >= 0 && graph_obj_ptr && graph_ptr && edge_ptr, GVQA_E_INVALID, "scene_graph_collate: null argument");
    GVQA_REQUIRE((N == 0 || (x_tokens && batch)) && (E == 0 || (edge_index && edge_tokens)) && (A == 0 || added_sym_edge), GVQA_E_INVALID,
                 "scene_graph_collate: null output");
    PairSet pairs;
    std::vector<int32_t> indeg((size_t)std::max<int64_t>(N, 0), 0);
    int64_t n_off = 0, e_off = 0, a_off = 0;
    graph_ptr[0] = 0; edge_ptr[0] = 0;
    auto put_edge = [&](int64_t src, int64_t dst, int64_t tok) {
        edge_index[e_off] = src; edge_index[E + e_off] = dst; edge_tokens[e_off] = tok;
        ++indeg[(size_t)dst];
        ++e_off;
    };
    for (int64_t g = 0; g < B; ++g) {
        int64_t n, e, a;
        int rc = graph_sizes(g, graph_obj_ptr, rel_ptr, rel_dst, n, e, a, pairs);
        if (rc) return rc;
        GVQA_REQUIRE(n_off + n <= N && e_off + e <= E && a_off + a <= A, GVQA_E_WORKSPACE, "scene_graph_collate: output sizes smaller than the batch");
        const int o0 = graph_obj_ptr[g], o1 = graph_obj_ptr[g + 1];
        for (int64_t i = 0; i < n; ++i) batch[n_off + i] = g;
        if (o1 == o0) {                               // empty scene graph: nodes "0" <-> "1", everything <UNK> (:196-224)
            for (int64_t i = 0; i < 2; ++i) {
                int64_t* xr = x_tokens + (n_off + i) * MAX_OBJ_TOKENS;
                for (int k = 0; k < MAX_OBJ_TOKENS; ++k) xr[k] = pad_tok;
                xr[0] = unk_tok; xr[1] = unk_tok;
                put_edge(n_off + i, n_off + i, self_tok);
                put_edge(n_off + i, n_off + 1 - i, unk_tok);
            }
        } else {
            for (int o = o0; o < o1; ++o) {
                const int64_t i = n_off + (o - o0);
                int64_t* xr = x_tokens + i * MAX_OBJ_TOKENS;
                for (int k = 0; k < MAX_OBJ_TOKENS; ++k) xr[k] = pad_tok;
                xr[0] = name_tok[o];
                const int na = attr_ptr[o + 1] - attr_ptr[o];
                GVQA_REQUIRE(na >= 0 && na < MAX_OBJ_TOKENS, GVQA_E_INVALID, "scene_graph_collate: object %d has %d distinct attributes (at most %d fit)", o, na,
                             MAX_OBJ_TOKENS - 1);
                for (int q = 0; q < na; ++q) xr[1 + q] = attr_tok[attr_ptr[o] + q];
                put_edge(i, i, self_tok);
                for (int r = rel_ptr[o]; r < rel_ptr[o + 1]; ++r) {
                    const int64_t j = n_off + rel_dst[r];
                    put_edge(i, j, rel_tok[r]);
                    if (!pairs.count(rel_dst[r], o - o0)) {
                        added_sym_edge[a_off++] = e_off;
                        put_edge(j, i, rel_tok[r]);   // the added reverse edge re-uses the relation's token (:327)
                    }
                }
            }
        }
        n_off += n;
        graph_ptr[g + 1] = (int32_t)n_off;
        edge_ptr[g + 1] = (int32_t)e_off;
    }
    GVQA_REQUIRE(n_off == N && e_off == E && a_off == A, GVQA_E_INVALID, "scene_graph_collate: sizes do not match the batch (call gvqa_scene_graph_collate_sizes)");
    int32_t md = 0;
    for (int32_t d : indeg) md = std::max(md, d);
    if (max_in_degree) *max_in_degree = md;
    return GVQA_OK;
}

}  // extern "C"
