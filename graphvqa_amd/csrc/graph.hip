// Destination-sorted CSR build for batched scene graphs (gfx950).
//
// Replaces the per-hop COO indexing PyG's MessagePassing.__collect__ / torch_scatter do for the
// reference (call site gat_skip.py:155-156; input contract gqa_dataset_entry.py:361-369,654).
// Integer work, HBM-bound and tiny (E ~ 1e5..1e6): a counting sort by destination with an
// in-row rank pass that orders every row by original edge id, so that per-node reductions run
// in the reference's COO order (deterministic, run-to-run bit-identical).
#include <algorithm>

#include <algorithm>
#include <vector>

#include "common.h"

namespace gvqa {

enum { ST_MAX_GNODES = 0, ST_MAX_GEDGES = 1, ST_MAX_DEG = 2, ST_NOT_INTRA = 3, ST_INVALID = 4 };

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// One pass over edges and nodes: in-degree histogram (+ arrival rank), int32 copy of `batch`,
// graph boundaries, contract validation.
__global__ __launch_bounds__(256) void k_count(int64_t N, int64_t E, int64_t B,
                                               const int64_t* __restrict__ edge_index,
                                               const int64_t* __restrict__ batch, int32_t* deg,
                                               int32_t* rank, int32_t* node_graph, int32_t* graph_ptr,
                                               int32_t* stats) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) {
        int64_t s = edge_index[i], d = edge_index[E + i];
        if (s < 0 || s >= N || d < 0 || d >= N) {
            stats[ST_INVALID] = 1;
            rank[i] = -1;
        } else {
            rank[i] = atomicAdd(&deg[d], 1);
            if (batch && batch[s] != batch[d]) stats[ST_NOT_INTRA] = 1;
        }
    }
    if (i < N) {
        int64_t g = batch ? batch[i] : 0;
        int64_t gprev = (i == 0) ? -1 : (batch ? batch[i - 1] : 0);
        // the fill loop below runs from the NEIGHBOUR's id: a malformed neighbour (negative sentinel, id >= B) must not
        // drive it out of graph_ptr[] or into a billion-iteration spin before finalize can report the batch
        if (g < 0 || g >= B || g < gprev || gprev < -1 || gprev >= B) {
            stats[ST_INVALID] = 1;
        } else {
            node_graph[i] = (int32_t)g;
            for (int64_t q = gprev + 1; q <= g; ++q) graph_ptr[q] = (int32_t)i;   // first node of graph q
            if (i == N - 1)
                for (int64_t q = g + 1; q <= B; ++q) graph_ptr[q] = (int32_t)N;
        }
    }
}

// ---- exclusive scan of deg[0..n) -> rowptr[0..n), three small kernels ------------------------
__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* lds /* >= 8 ints */) {
    // inclusive wave scan by shuffles, then cross-wave offsets through LDS (4 waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        int s = lds[w];
        if (w < wave) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(int64_t n, const int32_t* __restrict__ in,
                                                                 int32_t* tile_sum) {
    __shared__ int lds[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    int tot;
    (void)block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_offsets(int nt, int32_t* tile_sum) {
    __shared__ int lds[8];
    int carry = 0;
    for (int b0 = 0; b0 < nt; b0 += SCAN_THREADS) {
        int i = b0 + threadIdx.x;
        int v = i < nt ? tile_sum[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot, lds);
        if (i < nt) tile_sum[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(int64_t n, const int32_t* __restrict__ in,
                                                             const int32_t* __restrict__ tile_off,
                                                             int32_t* out) {
    __shared__ int lds[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot, lds) + tile_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

__global__ __launch_bounds__(256) void k_place(int64_t E, const int64_t* __restrict__ edge_index,
                                               const int32_t* __restrict__ rowptr,
                                               const int32_t* __restrict__ rank, int32_t* slot_eid) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int r = rank[e];
    if (r < 0) return;
    int64_t d = edge_index[E + e];
    slot_eid[rowptr[d] + r] = (int32_t)e;
}

// Rank sort inside each row: slot s holds an arbitrary arrival order; its final position is the
// number of edges of the same row with a smaller original id.  Rows are short (E/N ~ 2..5 for
// scene graphs), so the O(deg^2) row scan is cheaper than a general sort and needs no atomics.
__global__ __launch_bounds__(256) void k_rank_rows(int64_t N, int64_t E, const int64_t* __restrict__ edge_index,
                                                   const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ slot_eid,
                                                   int32_t* csr_eid, int32_t* csr_src) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= E) return;
    int32_t e = slot_eid[s];
    int64_t d = edge_index[E + e];
    if (d < 0 || d >= N) return;   // malformed input: flagged in k_count, reported by finalize
    int lo = rowptr[d], hi = rowptr[d + 1];
    int pos = lo;
    for (int t = lo; t < hi; ++t) pos += (slot_eid[t] < e) ? 1 : 0;
    csr_eid[pos] = e;
    csr_src[pos] = (int32_t)edge_index[e];
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// One atomic per wave and statistic (a per-thread atomicMax on three words serialises the chip).
__global__ __launch_bounds__(256) void k_stats(int64_t N, int64_t B, const int32_t* __restrict__ rowptr,
                                               const int32_t* __restrict__ graph_ptr, int32_t* stats) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int deg = 0, gn = 0, ge = 0;
    if (i < N) deg = rowptr[i + 1] - rowptr[i];
    if (i < B) {
        int n0 = graph_ptr[i], n1 = graph_ptr[i + 1];
        gn = n1 - n0;
        ge = rowptr[n1] - rowptr[n0];
    }
    deg = wave_max(deg); gn = wave_max(gn); ge = wave_max(ge);
    if ((threadIdx.x & 63) == 0) {
        if (deg > 0) atomicMax(&stats[ST_MAX_DEG], deg);
        if (gn > 0) atomicMax(&stats[ST_MAX_GNODES], gn);
        if (ge > 0) atomicMax(&stats[ST_MAX_GEDGES], ge);
    }
}

struct GraphLayout {
    size_t rowptr, csr_src, csr_eid, node_graph, graph_ptr, stats, row_group, row_group_e, deg, rank, slot_eid, tile_sum, graph_eptr, row_order, total;
    size_t clear_bytes;         // what gvqa_graph_build zeroes: everything in front of the packed plan's arrays (they are written in full when used)
    size_t pk_plan, pk_rowptr, pk_csr_src, pk_csr_eid, pk_node_graph, pk_node_old;      // packed row groups (plan_packed)
};

constexpr int ROW_GROUP = 128;      // rows of one group of the fused hop kernel (half a 256-row block tile)

// graph_eptr[g] = first CSR slot of graph g's in-edges (g = B: E) -- what the host needs beside graph_ptr to plan row groups
__global__ __launch_bounds__(256) void k_graph_eptr(int64_t B, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ rowptr,
                                                    int32_t* __restrict__ graph_eptr) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g <= B) graph_eptr[g] = rowptr[graph_ptr[g]];
}

// order[ns + r] = local index of the row with the r-th most in-edges of its group (ties in row order): the fused hop's edge
// loop runs to the largest in-degree among the rows a wave covers, so it walks the rows in this order.  One block per group.
__global__ __launch_bounds__(ROW_GROUP) void k_row_group_order(const int32_t* __restrict__ group_ptr, const int32_t* __restrict__ rowptr,
                                                               int32_t* __restrict__ order) {
    __shared__ int deg_s[ROW_GROUP];
    const int i = threadIdx.x;
    const int ns = group_ptr[blockIdx.x], cnt = group_ptr[blockIdx.x + 1] - ns;
    const int d = i < cnt ? rowptr[ns + i + 1] - rowptr[ns + i] : -1;
    deg_s[i] = d;
    __syncthreads();
    if (i >= cnt) return;
    int rank = 0;
    for (int j = 0; j < cnt; ++j) {
        const int dj = deg_s[j];
        rank += (dj > d || (dj == d && j < i)) ? 1 : 0;
    }
    order[ns + rank] = i;
}


// ---- grouped build: the whole CSR build of a loader-side layout in ONE launch --------------------------------------------
// When the loader vouches that the COO edges are grouped by graph (graph g's edges are COO positions [edge_ptr[g],
// edge_ptr[g + 1]) -- what Batch.from_data_list yields, gqa_dataset_entry.py:654), a row group's in-edges are a contiguous COO
// range, and one workgroup per row group does histogram, scan, placement, the in-row ordering by edge id and the row order out
// of LDS: one upload + one launch instead of a memset, seven kernels and an upload (28 -> 8 us of a step's stage time).
constexpr int GROUPED_EDGE_CAP = 8192;       // COO edges of one row group held in LDS (40 KiB)
__global__ __launch_bounds__(256) void k_build_grouped(int64_t N, int64_t E, int64_t B, const int64_t* __restrict__ edge_index,
                                                       const int64_t* __restrict__ batch, const int32_t* __restrict__ group_ptr,
                                                       const int32_t* __restrict__ group_eptr, const int32_t* __restrict__ graph_ptr,
                                                       int32_t* __restrict__ rowptr, int32_t* __restrict__ csr_src,
                                                       int32_t* __restrict__ csr_eid, int32_t* __restrict__ node_graph,
                                                       int32_t* __restrict__ order, int32_t* __restrict__ stats) {
    __shared__ int deg_s[ROW_GROUP], start_s[ROW_GROUP], cursor_s[ROW_GROUP], scan_s[8];
    __shared__ int gid_s[ROW_GROUP];          // graph id of the group's rows: a row group packs SEVERAL graphs
    __shared__ int eid_s[GROUPED_EDGE_CAP];
    __shared__ unsigned char row_s[GROUPED_EDGE_CAP];
    const int tid = threadIdx.x, r = blockIdx.x;
    const int ns = group_ptr[r], cnt = group_ptr[r + 1] - ns;
    const int e0 = group_eptr[r], ne = group_eptr[r + 1] - e0;
    if (tid < ROW_GROUP) {
        deg_s[tid] = 0;
        gid_s[tid] = (batch && tid < cnt) ? (int)batch[ns + tid] : 0;
    }
    __syncthreads();
    // an edge of this COO range must join two nodes of this group (the loader's promise); anything else is flagged and skipped
    auto edge_ok = [&](int64_t s, int64_t d) { return s >= ns && s < ns + cnt && d >= ns && d < ns + cnt; };
    for (int k = tid; k < ne; k += 256) {
        const int64_t s = edge_index[e0 + k], d = edge_index[E + e0 + k];
        if (edge_ok(s, d)) {
            atomicAdd(&deg_s[(int)d - ns], 1);
            // inside the group but across two of its graphs: a legal edge for the general kernels, yet this handle says
            // intra_graph = 1 and the fused hops rely on it -- flagged for gvqa_graph_check_valid (the edge is still placed)
            if (gid_s[(int)s - ns] != gid_s[(int)d - ns]) stats[ST_NOT_INTRA] = 1;
        } else {
            if (s >= 0 && s < N && d >= 0 && d < N && batch && batch[s] != batch[d]) stats[ST_NOT_INTRA] = 1;
            stats[ST_INVALID] = 1;
        }
    }
    __syncthreads();
    int total = 0;
    const int mydeg = tid < cnt ? deg_s[tid] : 0;
    const int ex = block_exclusive_scan(mydeg, &total, scan_s);
    if (tid < cnt) {
        start_s[tid] = ex;
        cursor_s[tid] = ex;
        rowptr[ns + tid] = e0 + ex;
        // node -> graph, checked against the uploaded layout
        const int64_t g = batch ? batch[ns + tid] : 0;
        if (g < 0 || g >= B || graph_ptr[g] > ns + tid || graph_ptr[g + 1] <= ns + tid) stats[ST_INVALID] = 1;
        else node_graph[ns + tid] = (int32_t)g;
    }
    if (tid == 0 && r == (int)gridDim.x - 1) rowptr[N] = (int32_t)E;
    if (total != ne && tid == 0) stats[ST_INVALID] = 1;
    __syncthreads();
    for (int k = tid; k < ne; k += 256) {
        const int64_t s = edge_index[e0 + k], d = edge_index[E + e0 + k];
        if (edge_ok(s, d)) {
            const int slot = atomicAdd(&cursor_s[(int)d - ns], 1);
            eid_s[slot] = e0 + k;
            row_s[slot] = (unsigned char)((int)d - ns);
        }
    }
    __syncthreads();
    // in-row order by original edge id (rows are short: the O(deg^2) count needs no sort), then the CSR arrays
    for (int slot = tid; slot < total; slot += 256) {
        const int e = eid_s[slot], row = row_s[slot];
        const int lo = start_s[row], hi = lo + deg_s[row];
        int pos = lo;
        for (int t = lo; t < hi; ++t) pos += eid_s[t] < e ? 1 : 0;
        csr_eid[e0 + pos] = e;
        csr_src[e0 + pos] = (int32_t)edge_index[e];
    }
    // rows of the group by in-degree, largest first (ties in row order): what k_row_group_order computes
    if (tid < cnt) {
        int rank = 0;
        for (int j = 0; j < cnt; ++j) {
            const int dj = deg_s[j];
            rank += (dj > mydeg || (dj == mydeg && j < tid)) ? 1 : 0;
        }
        order[ns + rank] = tid;
    }
}

static GraphLayout graph_layout(int64_t N, int64_t E, int64_t B) {
    GraphLayout L;
    size_t off = 0;
    auto take = [&](size_t count) {
        size_t r = off;
        off += align_up(count * sizeof(int32_t), 256);
        return r;
    };
    L.rowptr = take(N + 1);
    L.csr_src = take(E);
    L.csr_eid = take(E);
    L.node_graph = take(N);
    L.graph_ptr = take(B + 1);
    L.stats = take(8);
    L.row_group = take(B + 2);            // at most one group per non-empty graph, + the end marker
    L.row_group_e = take(B + 2);          // (grouped build: first COO edge of every row group)
    L.deg = take(N + 1);
    L.rank = take(E);
    L.slot_eid = take(E);
    L.tile_sum = take(cdiv(N + 1, SCAN_TILE) + 1);
    L.graph_eptr = take(B + 1);
    L.row_order = take(N);
    L.clear_bytes = off;
    // packed row groups: [group_ptr (B + 2) | group_eptr (B + 2) | group_gptr (B + 2) | graph_old (B)] uploaded as one image, then the packed CSR
    L.pk_plan = take(3 * (B + 2) + B);
    L.pk_rowptr = take(N + 1);
    L.pk_csr_src = take(E);
    L.pk_csr_eid = take(E);
    L.pk_node_graph = take(N);
    L.pk_node_old = take(N);
    L.total = off;
    return L;
}

// ---- packed row groups (gvqa_graph::pk_*) ---------------------------------------------------------------------------------
// The in-order groups below leave a ragged batch's groups partly empty; the aggregate-first hop (hopagg.hip) runs one workgroup
// per group and CU, so what counts is the NUMBER of groups against the CU count.  plan_packed re-orders the graphs (best fit
// decreasing: <= 128 nodes and <= PK_EDGE_CAP in-edges per group) and k_build_packed copies the CSR into that numbering: one
// block per packed group, rows copied slot for slot (same in-row COO order: every per-node sum is unchanged, bit for bit).
constexpr int PK_EDGE_CAP = 1024;       // = HA_ECAP of hopagg.hip: the CSR slice of a row group it keeps in LDS
__global__ __launch_bounds__(256) void k_build_packed(int64_t N, int64_t E, const int32_t* __restrict__ group_ptr, const int32_t* __restrict__ group_eptr,
                                                      const int32_t* __restrict__ group_gptr, const int32_t* __restrict__ graph_old,
                                                      const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ csr_src, const int32_t* __restrict__ csr_eid,
                                                      int32_t* __restrict__ pk_rowptr, int32_t* __restrict__ pk_csr_src, int32_t* __restrict__ pk_csr_eid,
                                                      int32_t* __restrict__ pk_node_graph, int32_t* __restrict__ pk_node_old) {
    __shared__ int gstart_s[ROW_GROUP + 1], gold0_s[ROW_GROUP], scan_s[8];
    const int tid = threadIdx.x, t = blockIdx.x;
    const int ns = group_ptr[t], cnt = group_ptr[t + 1] - ns;
    const int j0 = group_gptr[t], nj = min(group_gptr[t + 1] - j0, ROW_GROUP);
    const int e0 = group_eptr[t];
    if (tid < nj) {
        const int q = graph_old[j0 + tid];
        const int p0 = graph_ptr[q];
        gold0_s[tid] = p0;
        gstart_s[tid + 1] = graph_ptr[q + 1] - p0;          // (node count; prefix-summed below)
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int j = 0; j < nj; ++j) { const int c = gstart_s[j + 1]; gstart_s[j] = acc; acc += c; }
        gstart_s[nj] = acc;
    }
    __syncthreads();
    int j = 0, o = 0, lo = 0, deg = 0;
    if (tid < cnt) {
        while (j + 1 < nj && gstart_s[j + 1] <= tid) ++j;
        o = gold0_s[j] + (tid - gstart_s[j]);
        lo = rowptr[o];
        deg = rowptr[o + 1] - lo;
    }
    int total = 0;
    const int ex = block_exclusive_scan(deg, &total, scan_s);
    if (tid < cnt) {
        pk_rowptr[ns + tid] = e0 + ex;
        pk_node_old[ns + tid] = o;
        pk_node_graph[ns + tid] = j0 + j;
        const int shift = ns + gstart_s[j] - gold0_s[j];     // old node id -> packed node id inside this graph
        for (int k = 0; k < deg; ++k) {
            pk_csr_src[e0 + ex + k] = csr_src[lo + k] + shift;
            pk_csr_eid[e0 + ex + k] = csr_eid[lo + k];
        }
    }
    if (tid == 0 && t == (int)gridDim.x - 1) pk_rowptr[N] = (int32_t)E;
}

// Host side: best fit decreasing over the non-empty graphs (sizes are small integers: counting sort by node count, open groups
// bucketed by their free rows -- O(B x 128) worst case, O(B) in practice), then one upload and one launch.  `hp` / `he`: host copies
// of graph_ptr [B + 1] and of the graphs' first in-edge slots [B + 1] (edges counted by destination graph).
static int plan_packed(gvqa_graph* g, const int32_t* hp, const int32_t* he, hipStream_t stream) {
    g->pk_num_row_groups = 0;
    g->pk_max_row_group_edges = 0;
    g->pk_row_group_ptr = g->pk_rowptr = g->pk_csr_src = g->pk_csr_eid = g->pk_node_graph = g->pk_node_old = g->pk_graph_old = nullptr;
    const int mode = get_option(GVQA_OPT_PACKED_GROUPS);
    const int64_t B = g->num_graphs, N = g->num_nodes, E = g->num_edges;
    const int G0 = g->num_row_groups;
    if (mode == 0 || G0 <= 1 || !g->intra_graph || N <= 0 || B <= 0) return GVQA_OK;
    const int64_t cus = device_cu_count();
    const int64_t gmin = cdiv(N, ROW_GROUP);
    if (mode == 1 ? cdiv(gmin, cus) >= cdiv((int64_t)G0, cus) : gmin >= G0) return GVQA_OK;      // no order of the graphs can pay
    static thread_local std::vector<int32_t> order, bin_rows, bin_edges, bin_of, byrem_head, byrem_next;
    // graphs by node count, largest first (ties in batch order); empty graphs take no rows
    int32_t cnt_by_size[ROW_GROUP + 2] = {0};
    for (int64_t q = 0; q < B; ++q) {
        const int32_t n = hp[q + 1] - hp[q];
        if (n > ROW_GROUP || he[q + 1] - he[q] > PK_EDGE_CAP) return GVQA_OK;              // (a graph the aggregate-first hop cannot take anyway)
        ++cnt_by_size[n];
    }
    int32_t first_of_size[ROW_GROUP + 2];
    { int32_t acc = 0; for (int n = ROW_GROUP; n >= 0; --n) { first_of_size[n] = acc; acc += cnt_by_size[n]; } }
    order.resize(B);
    for (int64_t q = 0; q < B; ++q) order[first_of_size[hp[q + 1] - hp[q]]++] = (int32_t)q;
    const int64_t nonempty = B - cnt_by_size[0];
    bin_rows.clear(); bin_edges.clear();
    bin_of.assign(B, -1);
    byrem_head.assign(ROW_GROUP + 1, -1);       // open groups by free rows: singly linked stacks (head per free-row count, next per group)
    byrem_next.clear();
    for (int64_t k = 0; k < nonempty; ++k) {
        const int32_t q = order[k], n = hp[q + 1] - hp[q], e = he[q + 1] - he[q];
        int32_t take = -1, take_rem = 0, take_prev = -1;
        for (int rem = n; rem <= ROW_GROUP && take < 0; ++rem) {                           // best fit: the fullest group that still takes it
            int32_t prev = -1;
            for (int32_t b = byrem_head[rem]; b >= 0; prev = b, b = byrem_next[b])
                if (bin_edges[b] + e <= PK_EDGE_CAP) { take = b; take_rem = rem; take_prev = prev; break; }
        }
        if (take < 0) {
            take = (int32_t)bin_rows.size();
            bin_rows.push_back(0); bin_edges.push_back(0); byrem_next.push_back(-1);
        } else {                                                                           // unlink from its free-row list
            if (take_prev < 0) byrem_head[take_rem] = byrem_next[take];
            else byrem_next[take_prev] = byrem_next[take];
        }
        bin_rows[take] += n; bin_edges[take] += e;
        bin_of[q] = take;
        const int rem = ROW_GROUP - bin_rows[take];
        byrem_next[take] = byrem_head[rem];
        byrem_head[rem] = take;
    }
    const int64_t G1 = (int64_t)bin_rows.size();
    if (mode == 1 ? cdiv(G1, cus) >= cdiv((int64_t)G0, cus) : G1 >= G0) return GVQA_OK;
    // the upload image: [group_ptr | group_eptr | group_gptr | graph_old]; graphs of a group in placement order (largest first),
    // the empty graphs behind the last group
    GraphLayout L = graph_layout(N, E, B);
    char* base = const_cast<char*>(reinterpret_cast<const char*>(g->rowptr)) - L.rowptr;
    const size_t words = (size_t)(3 * (B + 2) + B);
    struct Slot { int32_t* v = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; int device = -1; };
    static thread_local Slot ring[4];
    static thread_local int next = 0;
    Slot& sl = ring[next];
    next = (next + 1) & 3;
    int dev = -1;
    GVQA_HIP_CHECK(hipGetDevice(&dev));
    if (sl.used) GVQA_HIP_CHECK(hipEventSynchronize(sl.done));
    if (sl.done && sl.device != dev) { GVQA_HIP_CHECK(hipEventDestroy(sl.done)); sl.done = nullptr; }
    if (!sl.done) { GVQA_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)); sl.device = dev; }
    if (sl.cap < words) {
        if (sl.v) GVQA_HIP_CHECK(hipHostFree(sl.v));
        sl.v = nullptr; sl.cap = 0;
        const size_t want = std::max<size_t>(words, 4096) * 2;
        GVQA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.v), want * sizeof(int32_t), hipHostMallocDefault));
        sl.cap = want;
    }
    int32_t *s_grp = sl.v, *s_gre = sl.v + (B + 2), *s_gg = sl.v + 2 * (B + 2), *s_old = sl.v + 3 * (B + 2);
    // group_gptr from the groups' graph counts, then the graphs dealt into their group's range in placement order
    for (int64_t b = 0; b <= G1; ++b) s_gg[b] = 0;
    for (int64_t k = 0; k < nonempty; ++k) ++s_gg[bin_of[order[k]] + 1];
    for (int64_t b = 0; b < G1; ++b) s_gg[b + 1] += s_gg[b];
    {
        static thread_local std::vector<int32_t> cursor;
        cursor.assign(s_gg, s_gg + G1);
        for (int64_t k = 0; k < nonempty; ++k) s_old[cursor[bin_of[order[k]]]++] = order[k];
        for (int64_t k = nonempty; k < B; ++k) s_old[k] = order[k];
    }
    int32_t max_e = 0;
    s_grp[0] = 0; s_gre[0] = 0;
    for (int64_t b = 0; b < G1; ++b) {
        s_grp[b + 1] = s_grp[b] + bin_rows[b];
        s_gre[b + 1] = s_gre[b] + bin_edges[b];
        max_e = std::max(max_e, bin_edges[b]);
    }
    int32_t* plan_dev = reinterpret_cast<int32_t*>(base + L.pk_plan);
    GVQA_HIP_CHECK(hipMemcpyAsync(plan_dev, sl.v, words * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    GVQA_HIP_CHECK(hipEventRecord(sl.done, stream));
    sl.used = true;
    auto P = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };
    hipLaunchKernelGGL(k_build_packed, dim3((unsigned)G1), dim3(256), 0, stream, N, E, plan_dev, plan_dev + (B + 2), plan_dev + 2 * (B + 2),
                       plan_dev + 3 * (B + 2), g->graph_ptr, g->rowptr, g->csr_src, g->csr_eid, P(L.pk_rowptr), P(L.pk_csr_src), P(L.pk_csr_eid),
                       P(L.pk_node_graph), P(L.pk_node_old));
    GVQA_LAUNCH_CHECK();
    g->pk_num_row_groups = (int32_t)G1;
    g->pk_max_row_group_edges = max_e;
    g->pk_row_group_ptr = plan_dev;
    g->pk_graph_old = plan_dev + 3 * (B + 2);
    g->pk_rowptr = P(L.pk_rowptr); g->pk_csr_src = P(L.pk_csr_src); g->pk_csr_eid = P(L.pk_csr_eid);
    g->pk_node_graph = P(L.pk_node_graph); g->pk_node_old = P(L.pk_node_old);
    return GVQA_OK;
}

// Row groups for the fused hop kernel: greedy in order, a group closes when the next graph would not fit.  `hp` / `he`:
// host copies of graph_ptr [B+1] and of the graphs' first in-edge slots [B+1].  The plan is uploaded asynchronously from a
// small ring of host buffers (a slot is reused only after its upload has completed).
// The cut rule both planners share.  A group closes before graph q when q's nodes would not fit any more, or when the group's
// graph IDS would span more than ROW_GROUP: the hop kernels keep per-graph state of a group in LDS arrays indexed by
// node_graph[row] - node_graph[first row] (scales, output maxima: 128 entries), and empty graphs between two non-empty ones
// take ids without taking rows.  `gfirst`: the first non-empty graph of the open group, -1 while it has no rows.
static inline bool row_group_closes_before(const int32_t* hp, int64_t q, int32_t start, int64_t& gfirst) {
    const int32_t n = hp[q + 1] - hp[q];
    const bool close = (hp[q + 1] - start > ROW_GROUP) || (n > 0 && gfirst >= 0 && q - gfirst >= ROW_GROUP);
    if (close) gfirst = -1;
    if (n > 0 && gfirst < 0) gfirst = q;
    return close;
}

static int plan_row_groups(gvqa_graph* g, const int32_t* hp, const int32_t* he, hipStream_t stream) {
    // PINNED host buffers: the upload is a true asynchronous DMA (from pageable memory the runtime stages the copy on the
    // calling thread -- 50 us of a 256-graph shard's 450 us step)
    struct Slot { int32_t* v = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; int device = -1; };
    static thread_local Slot ring[4];
    static thread_local int next = 0;
    Slot& sl = ring[next];
    next = (next + 1) & 3;
    int dev = -1;
    GVQA_HIP_CHECK(hipGetDevice(&dev));
    if (sl.used) GVQA_HIP_CHECK(hipEventSynchronize(sl.done));
    if (sl.done && sl.device != dev) {                // one host thread driving several GPUs: an event belongs to the device it was created on
        GVQA_HIP_CHECK(hipEventDestroy(sl.done));
        sl.done = nullptr;
    }
    if (!sl.done) { GVQA_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)); sl.device = dev; }
    const int64_t B = g->num_graphs, N = g->num_nodes;
    if (sl.cap < (size_t)B + 2) {
        if (sl.v) GVQA_HIP_CHECK(hipHostFree(sl.v));
        sl.v = nullptr; sl.cap = 0;
        const size_t want = std::max<size_t>((size_t)B + 2, 1024) * 2;
        GVQA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.v), want * sizeof(int32_t), hipHostMallocDefault));
        sl.cap = want;
    }
    struct { int32_t* p; size_t n; void push_back(int32_t x) { p[n++] = x; } size_t size() const { return n; } int32_t* data() { return p; } }
        hg{sl.v, 0};
    hg.push_back(0);
    int32_t start = 0, e_start = 0, max_e = 0;
    int64_t gfirst = -1;
    for (int64_t q = 0; q < B; ++q) {
        if (row_group_closes_before(hp, q, start, gfirst)) {
            max_e = std::max(max_e, he[q] - e_start);
            hg.push_back(hp[q]);
            start = hp[q];
            e_start = he[q];
        }
    }
    max_e = std::max(max_e, he[B] - e_start);
    hg.push_back((int32_t)N);
    GraphLayout L = graph_layout(N, g->num_edges, B);
    char* base = const_cast<char*>(reinterpret_cast<const char*>(g->rowptr)) - L.rowptr;
    int32_t* grp_dev = reinterpret_cast<int32_t*>(base + L.row_group);
    GVQA_HIP_CHECK(hipMemcpyAsync(grp_dev, hg.data(), hg.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    GVQA_HIP_CHECK(hipEventRecord(sl.done, stream));
    sl.used = true;
    g->row_group_ptr = grp_dev;
    g->num_row_groups = (int32_t)hg.size() - 1;
    g->max_row_group_edges = max_e;
    int32_t* order = reinterpret_cast<int32_t*>(base + L.row_order);
    hipLaunchKernelGGL(k_row_group_order, dim3((unsigned)g->num_row_groups), dim3(ROW_GROUP), 0, stream, grp_dev, g->rowptr, order);
    GVQA_LAUNCH_CHECK();
    g->row_group_order = order;
    return plan_packed(g, hp, he, stream);
}

}  // namespace gvqa

extern "C" {

size_t gvqa_graph_workspace_bytes(int64_t N, int64_t E, int64_t B) {
    if (N < 0 || E < 0 || B < 0) return 0;
    return gvqa::graph_layout(N, E, B).total;
}

int gvqa_graph_build(int64_t N, int64_t E, int64_t B, const int64_t* edge_index, const int64_t* batch,
                     void* ws, size_t ws_bytes, void* stream_, gvqa_graph* out) {
    using namespace gvqa;
    GVQA_REQUIRE(out, GVQA_E_INVALID, "gvqa_graph_build: null output handle");
    GVQA_REQUIRE(N >= 0 && E >= 0 && B >= 0, GVQA_E_INVALID, "gvqa_graph_build: negative size");
    GVQA_REQUIRE(N < (1ll << 31) - 1 && E < (1ll << 31) - 1, GVQA_E_INVALID,
                 "gvqa_graph_build: N and E must fit int32");
    GVQA_REQUIRE(E == 0 || edge_index, GVQA_E_INVALID, "gvqa_graph_build: null edge_index");
    GVQA_REQUIRE(batch || B == 1 || N == 0, GVQA_E_INVALID,
                 "gvqa_graph_build: batch == NULL requires num_graphs == 1");
    GraphLayout L = graph_layout(N, E, B);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE,
                 "gvqa_graph_build: workspace %zu < required %zu", ws_bytes, L.total);
    GVQA_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, GVQA_E_INVALID,
                 "gvqa_graph_build: workspace must be 256-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer timer(GVQA_STAGE_GRAPH, stream);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };

    GVQA_HIP_CHECK(hipMemsetAsync(ws, 0, L.clear_bytes, stream));
    int64_t m = N > E ? N : E;
    if (m > 0) {
        hipLaunchKernelGGL(k_count, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, stream, N, E, B, edge_index,
                           batch, P(L.deg), P(L.rank), P(L.node_graph), P(L.graph_ptr), P(L.stats));
        GVQA_LAUNCH_CHECK();
    }
    int nt = (int)cdiv(N + 1, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(nt), dim3(SCAN_THREADS), 0, stream, N + 1, P(L.deg), P(L.tile_sum));
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, stream, nt, P(L.tile_sum));
    hipLaunchKernelGGL(k_scan_apply, dim3(nt), dim3(SCAN_THREADS), 0, stream, N + 1, P(L.deg), P(L.tile_sum),
                       P(L.rowptr));
    GVQA_LAUNCH_CHECK();
    if (E > 0) {
        hipLaunchKernelGGL(k_place, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, stream, E, edge_index,
                           P(L.rowptr), P(L.rank), P(L.slot_eid));
        hipLaunchKernelGGL(k_rank_rows, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, stream, N, E, edge_index,
                           P(L.rowptr), P(L.slot_eid), P(L.csr_eid), P(L.csr_src));
        GVQA_LAUNCH_CHECK();
    }
    // (the size statistics -- k_stats -- are computed by gvqa_graph_finalize, the only reader: gvqa_graph_finalize_host takes
    //  them from the caller's per-graph layout instead and the step saves the launch)
    memset(out, 0, sizeof(*out));
    out->num_nodes = N;
    out->num_edges = E;
    out->num_graphs = B;
    out->rowptr = P(L.rowptr);
    out->csr_src = P(L.csr_src);
    out->csr_eid = P(L.csr_eid);
    out->node_graph = P(L.node_graph);
    out->graph_ptr = P(L.graph_ptr);
    out->stats_dev = P(L.stats);
    return GVQA_OK;
}

int gvqa_graph_finalize(gvqa_graph* g, void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(g && g->stats_dev, GVQA_E_INVALID, "gvqa_graph_finalize: graph not built");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int32_t st[8];
    {
        const int64_t mb = g->num_nodes > g->num_graphs ? g->num_nodes : g->num_graphs;
        if (mb > 0) {
            hipLaunchKernelGGL(k_stats, dim3((unsigned)cdiv(mb, 256)), dim3(256), 0, stream, g->num_nodes, g->num_graphs, g->rowptr,
                               g->graph_ptr, const_cast<int32_t*>(g->stats_dev));
            GVQA_LAUNCH_CHECK();
        }
    }
    GVQA_HIP_CHECK(hipMemcpyAsync(st, g->stats_dev, sizeof(st), hipMemcpyDeviceToHost, stream));
    GVQA_HIP_CHECK(hipStreamSynchronize(stream));
    g->max_graph_nodes = st[ST_MAX_GNODES];
    g->max_graph_edges = st[ST_MAX_GEDGES];
    g->max_in_degree = st[ST_MAX_DEG];
    g->intra_graph = st[ST_NOT_INTRA] ? 0 : 1;
    g->valid = st[ST_INVALID] ? 0 : 1;
    g->finalized = 1;
    g->row_group_ptr = nullptr;
    g->row_group_order = nullptr;
    g->num_row_groups = 0;
    g->max_row_group_edges = 0;
    GVQA_REQUIRE(g->valid, GVQA_E_GRAPH,
                 "graph violates the input contract (edge index out of [0,N), or batch not "
                 "non-decreasing in [0,B))");
    // Row groups for the fused hop kernel, planned on the host from graph_ptr / the graphs' edge offsets (two small
    // copies; this call synchronises anyway).
    const int64_t B = g->num_graphs, N = g->num_nodes;
    if (g->intra_graph && N > 0 && B > 0 && g->max_graph_nodes <= ROW_GROUP && B < (1ll << 24)) {
        GraphLayout L = graph_layout(N, g->num_edges, B);
        char* base = const_cast<char*>(reinterpret_cast<const char*>(g->rowptr)) - L.rowptr;
        int32_t* eptr_dev = reinterpret_cast<int32_t*>(base + L.graph_eptr);
        hipLaunchKernelGGL(k_graph_eptr, dim3((unsigned)cdiv(B + 1, 256)), dim3(256), 0, stream, B, g->graph_ptr, g->rowptr, eptr_dev);
        GVQA_LAUNCH_CHECK();
        static thread_local std::vector<int32_t> hp, he;
        hp.resize(B + 1); he.resize(B + 1);
        GVQA_HIP_CHECK(hipMemcpyAsync(hp.data(), g->graph_ptr, (B + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        GVQA_HIP_CHECK(hipMemcpyAsync(he.data(), eptr_dev, (B + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        GVQA_HIP_CHECK(hipStreamSynchronize(stream));
        return plan_row_groups(g, hp.data(), he.data(), stream);
    }
    return GVQA_OK;
}

int gvqa_graph_finalize_host(gvqa_graph* g, const int32_t* graph_ptr_host, const int32_t* graph_edge_ptr_host, int32_t max_in_degree,
                             void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(g && g->stats_dev, GVQA_E_INVALID, "gvqa_graph_finalize_host: graph not built");
    GVQA_REQUIRE(graph_ptr_host && graph_edge_ptr_host, GVQA_E_INVALID, "gvqa_graph_finalize_host: null layout");
    const int64_t B = g->num_graphs, N = g->num_nodes, E = g->num_edges;
    GVQA_REQUIRE(graph_ptr_host[0] == 0 && graph_ptr_host[B] == N && graph_edge_ptr_host[0] == 0 && graph_edge_ptr_host[B] == E,
                 GVQA_E_GRAPH, "gvqa_graph_finalize_host: layout does not span the batch (%d..%d nodes, %d..%d edges)",
                 graph_ptr_host[0], graph_ptr_host[B], graph_edge_ptr_host[0], graph_edge_ptr_host[B]);
    int32_t mn = 0, me = 0;
    for (int64_t q = 0; q < B; ++q) {
        const int32_t n = graph_ptr_host[q + 1] - graph_ptr_host[q], e = graph_edge_ptr_host[q + 1] - graph_edge_ptr_host[q];
        GVQA_REQUIRE(n >= 0 && e >= 0, GVQA_E_GRAPH, "gvqa_graph_finalize_host: layout not monotone at graph %lld", (long long)q);
        mn = std::max(mn, n);
        me = std::max(me, e);
    }
    g->max_graph_nodes = mn;
    g->max_graph_edges = me;
    g->max_in_degree = max_in_degree > 0 ? max_in_degree : me;      // unknown: the bound that is always true
    g->intra_graph = 1;         // the loader's promise (what `Batch.from_data_list` produces, gqa_dataset_entry.py:654)
    g->valid = 1;
    g->finalized = 1;
    g->row_group_ptr = nullptr;
    g->row_group_order = nullptr;
    g->num_row_groups = 0;
    g->max_row_group_edges = 0;
    if (N > 0 && B > 0 && mn <= ROW_GROUP && B < (1ll << 24))
        return plan_row_groups(g, graph_ptr_host, graph_edge_ptr_host, static_cast<hipStream_t>(stream_));
    return GVQA_OK;
}

int gvqa_graph_build_grouped(int64_t N, int64_t E, int64_t B, const int64_t* edge_index, const int64_t* batch,
                             const int32_t* graph_ptr_host, const int32_t* graph_edge_ptr_host, int32_t max_in_degree,
                             void* ws, size_t ws_bytes, void* stream_, gvqa_graph* out) {
    using namespace gvqa;
    GVQA_REQUIRE(out && graph_ptr_host && graph_edge_ptr_host, GVQA_E_INVALID, "gvqa_graph_build_grouped: null argument");
    GVQA_REQUIRE(N >= 0 && E >= 0 && B >= 0 && N < (1ll << 31) - 1 && E < (1ll << 31) - 1, GVQA_E_INVALID, "gvqa_graph_build_grouped: bad size");
    if (N == 0 || E == 0 || B == 0 || B >= (1ll << 24) || !edge_index || (!batch && B != 1)) return GVQA_E_UNSUPPORTED;   // (quietly: the caller takes the general pair)
    GraphLayout L = graph_layout(N, E, B);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE, "gvqa_graph_build_grouped: workspace %zu < required %zu", ws_bytes, L.total);
    GVQA_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, GVQA_E_INVALID, "gvqa_graph_build_grouped: workspace must be 256-byte aligned");
    const int32_t *hp = graph_ptr_host, *he = graph_edge_ptr_host;
    GVQA_REQUIRE(hp[0] == 0 && hp[B] == N && he[0] == 0 && he[B] == E, GVQA_E_GRAPH,
                 "gvqa_graph_build_grouped: layout does not span the batch (%d..%d nodes, %d..%d edges)", hp[0], hp[B], he[0], he[B]);
    int32_t mn = 0, me = 0;
    for (int64_t q = 0; q < B; ++q) {
        const int32_t n = hp[q + 1] - hp[q], e = he[q + 1] - he[q];
        GVQA_REQUIRE(n >= 0 && e >= 0, GVQA_E_GRAPH, "gvqa_graph_build_grouped: layout not monotone at graph %lld", (long long)q);
        mn = std::max(mn, n);
        me = std::max(me, e);
    }
    if (mn > ROW_GROUP) return GVQA_E_UNSUPPORTED;
    // staging image of the contiguous device range [graph_ptr | stats | row_group | row_group_e] (pinned ring, as plan_row_groups)
    const size_t span = L.row_group_e + align_up((size_t)(B + 2) * sizeof(int32_t), 256) - L.graph_ptr;
    struct Slot { char* v = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; int device = -1; };
    static thread_local Slot ring[4];
    static thread_local int next = 0;
    Slot& sl = ring[next];
    int dev = -1;
    GVQA_HIP_CHECK(hipGetDevice(&dev));
    if (sl.used) GVQA_HIP_CHECK(hipEventSynchronize(sl.done));
    if (sl.done && sl.device != dev) { GVQA_HIP_CHECK(hipEventDestroy(sl.done)); sl.done = nullptr; }
    if (!sl.done) { GVQA_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)); sl.device = dev; }
    if (sl.cap < span) {
        if (sl.v) GVQA_HIP_CHECK(hipHostFree(sl.v));
        sl.v = nullptr; sl.cap = 0;
        const size_t want = std::max<size_t>(span, 16384) * 2;
        GVQA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.v), want, hipHostMallocDefault));
        sl.cap = want;
    }
    memset(sl.v, 0, span);
    int32_t* s_gp = reinterpret_cast<int32_t*>(sl.v);
    int32_t* s_grp = reinterpret_cast<int32_t*>(sl.v + (L.row_group - L.graph_ptr));
    int32_t* s_gre = reinterpret_cast<int32_t*>(sl.v + (L.row_group_e - L.graph_ptr));
    memcpy(s_gp, hp, (size_t)(B + 1) * sizeof(int32_t));
    int G = 0;
    int32_t start = 0, e_start = 0, max_e = 0;
    s_grp[0] = 0; s_gre[0] = 0;
    int64_t gfirst = -1;
    for (int64_t q = 0; q < B; ++q) {
        if (row_group_closes_before(hp, q, start, gfirst)) {
            max_e = std::max(max_e, he[q] - e_start);
            ++G;
            s_grp[G] = hp[q]; s_gre[G] = he[q];
            start = hp[q]; e_start = he[q];
        }
    }
    max_e = std::max(max_e, he[B] - e_start);
    ++G;
    s_grp[G] = (int32_t)N; s_gre[G] = (int32_t)E;
    if (max_e > GROUPED_EDGE_CAP) return GVQA_E_UNSUPPORTED;      // (nothing enqueued yet; the slot is not consumed)
    next = (next + 1) & 3;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer timer(GVQA_STAGE_GRAPH, stream);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };
    GVQA_HIP_CHECK(hipMemcpyAsync(base + L.graph_ptr, sl.v, span, hipMemcpyHostToDevice, stream));
    GVQA_HIP_CHECK(hipEventRecord(sl.done, stream));
    sl.used = true;
    hipLaunchKernelGGL(k_build_grouped, dim3((unsigned)G), dim3(256), 0, stream, N, E, B, edge_index, batch, P(L.row_group), P(L.row_group_e),
                       P(L.graph_ptr), P(L.rowptr), P(L.csr_src), P(L.csr_eid), P(L.node_graph), P(L.row_order), P(L.stats));
    GVQA_LAUNCH_CHECK();
    memset(out, 0, sizeof(*out));
    out->num_nodes = N; out->num_edges = E; out->num_graphs = B;
    out->rowptr = P(L.rowptr); out->csr_src = P(L.csr_src); out->csr_eid = P(L.csr_eid);
    out->node_graph = P(L.node_graph); out->graph_ptr = P(L.graph_ptr); out->stats_dev = P(L.stats);
    out->max_graph_nodes = mn; out->max_graph_edges = me;
    out->max_in_degree = max_in_degree > 0 ? max_in_degree : me;
    out->intra_graph = 1; out->valid = 1; out->finalized = 1;
    out->row_group_ptr = P(L.row_group); out->num_row_groups = G; out->max_row_group_edges = max_e;
    out->row_group_order = P(L.row_order);
    return plan_packed(out, hp, he, stream);
}

// The deferred check behind gvqa_graph_finalize_host: that call reads nothing back, so a wrong loader-side layout (counts by
// source on a batch with cross-graph edges, stale counts, out-of-range ids) goes unnoticed and the kernels that size their LDS
// regions from it would work on wrong bounds.  This call synchronises: it reads the validation flags the build left on the device
// and the per-graph node / in-edge layout the DEVICE derives from the arrays, and compares them with what the handle holds.
int gvqa_graph_check_valid(const gvqa_graph* g, void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(g && g->stats_dev && g->finalized, GVQA_E_INVALID, "gvqa_graph_check_valid: graph not built / finalized");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t B = g->num_graphs, N = g->num_nodes;
    const int64_t mb = N > B ? N : B;
    if (mb > 0) {
        hipLaunchKernelGGL(k_stats, dim3((unsigned)cdiv(mb, 256)), dim3(256), 0, stream, N, B, g->rowptr, g->graph_ptr,
                           const_cast<int32_t*>(g->stats_dev));
        GVQA_LAUNCH_CHECK();
    }
    int32_t st[8];
    GVQA_HIP_CHECK(hipMemcpyAsync(st, g->stats_dev, sizeof(st), hipMemcpyDeviceToHost, stream));
    GVQA_HIP_CHECK(hipStreamSynchronize(stream));
    GVQA_REQUIRE(!st[ST_INVALID], GVQA_E_GRAPH, "graph violates the input contract (edge index out of [0,N), or batch not non-decreasing in [0,B))");
    GVQA_REQUIRE(!(st[ST_NOT_INTRA] && g->intra_graph), GVQA_E_GRAPH, "graph handle claims an intra-graph batch, but an edge joins two graphs");
    GVQA_REQUIRE(st[ST_MAX_GNODES] <= g->max_graph_nodes && st[ST_MAX_GEDGES] <= g->max_graph_edges && st[ST_MAX_DEG] <= g->max_in_degree,
                 GVQA_E_GRAPH, "graph handle's statistics are below the batch's (largest graph %d nodes / %d in-edges, in-degree %d; handle: %d / %d / %d)",
                 st[ST_MAX_GNODES], st[ST_MAX_GEDGES], st[ST_MAX_DEG], g->max_graph_nodes, g->max_graph_edges, g->max_in_degree);
    return GVQA_OK;
}

// ---- per-graph rows <-> node rows (training glue: the instruction halves of the concatenations) -----------
// out[i, :] = rows[node_graph[i], :]  and its adjoint  out[b, :] = sum_{i in graph b} x[i, :]  (nodes of a graph
// are contiguous: graph_ptr).  Deterministic; one thread per column, rows in order.
namespace gvqa {
__global__ __launch_bounds__(256) void k_graph_rows_to_nodes(int64_t N, int F, const int32_t* __restrict__ node_graph,
                                                             const float* __restrict__ rows, int64_t ldr, float* __restrict__ out,
                                                             int64_t ldo, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F) return;
    for (int64_t i = blockIdx.y; i < N; i += gridDim.y) {
        const float v = rows[(int64_t)node_graph[i] * ldr + c];
        float* o = out + i * ldo + c;
        *o = accumulate ? *o + v : v;
    }
}
__global__ __launch_bounds__(256) void k_graph_segment_sum(int F, const int32_t* __restrict__ graph_ptr, const float* __restrict__ x,
                                                           int64_t ldx, float* __restrict__ out, int64_t ldo, int mean) {
    const int c = blockIdx.y * 256 + threadIdx.x, b = blockIdx.x;      // graphs on grid.x (no 65535 limit)
    if (c >= F) return;
    float acc = 0.f;
    const int lo = graph_ptr[b], hi = graph_ptr[b + 1];
    for (int i = lo; i < hi; ++i) acc += x[(int64_t)i * ldx + c];
    out[(int64_t)b * ldo + c] = mean ? acc / (float)max(hi - lo, 1) : acc;      // (scatter_mean: divides by max(count, 1))
}
}  // namespace gvqa

// out[i, :F] = sum over the CSR row of node i of x[csr_eid[s], :F]  (rows of a per-edge tensor summed per destination
// node -- or per SOURCE node on the transposed graph): the adjoint of the per-edge row gathers x[dst] / x[src], and
// torch_scatter's scatter_add by destination.  One wave per node, rows added in slot order (deterministic).
namespace gvqa {
__global__ __launch_bounds__(256) void k_csr_row_sum(int N, int F, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_eid,
                                                     const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    for (int c = lane; c < F; c += 64) {
        float acc = 0.f;
        for (int s = lo; s < hi; ++s) acc += x[(int64_t)csr_eid[s] * ldx + c];
        out[(int64_t)i * ldo + c] = acc;
    }
}
}  // namespace gvqa

int gvqa_graph_edge_rows_sum(const gvqa_graph* g, int64_t F, const float* x, int64_t ld_x, float* out, int64_t ld_out, void* stream) {
    using namespace gvqa;
    GVQA_REQUIRE(g && g->valid, GVQA_E_INVALID, "graph_edge_rows_sum: graph not built");
    GVQA_REQUIRE(F >= 0 && F < (1ll << 31) && ld_x >= F && ld_out >= F, GVQA_E_INVALID, "graph_edge_rows_sum: bad sizes");
    if (g->num_nodes == 0 || F == 0) return GVQA_OK;
    GVQA_REQUIRE((x || g->num_edges == 0) && out, GVQA_E_INVALID, "graph_edge_rows_sum: null tensor");
    hipLaunchKernelGGL(k_csr_row_sum, dim3((unsigned)cdiv(g->num_nodes, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (int)g->num_nodes, (int)F, g->rowptr, g->csr_eid, x, ld_x, out, ld_out);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

int gvqa_graph_rows_to_nodes(const gvqa_graph* g, int64_t F, const float* rows, int64_t ld_rows, float* out, int64_t ld_out,
                             int accumulate, void* stream) {
    using namespace gvqa;
    GVQA_REQUIRE(g && g->valid, GVQA_E_INVALID, "graph_rows_to_nodes: graph not built");
    GVQA_REQUIRE(F >= 0 && F < (1ll << 31) && ld_rows >= F && ld_out >= F, GVQA_E_INVALID, "graph_rows_to_nodes: bad sizes");
    if (g->num_nodes == 0 || F == 0) return GVQA_OK;
    GVQA_REQUIRE(rows && out, GVQA_E_INVALID, "graph_rows_to_nodes: null tensor");
    const dim3 grid((unsigned)cdiv(F, 256), (unsigned)std::min<int64_t>(g->num_nodes, 8192));
    hipLaunchKernelGGL(k_graph_rows_to_nodes, grid, dim3(256), 0, static_cast<hipStream_t>(stream), g->num_nodes, (int)F,
                       g->node_graph, rows, ld_rows, out, ld_out, accumulate);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

namespace gvqa {
static int graph_segment_reduce(const gvqa_graph* g, int64_t F, const float* x, int64_t ld_x, float* out, int64_t ld_out, int mean,
                                void* stream) {
    GVQA_REQUIRE(g && g->valid, GVQA_E_INVALID, "graph_segment_sum: graph not built");
    GVQA_REQUIRE(F >= 0 && F < 65535ll * 256 && ld_x >= F && ld_out >= F, GVQA_E_INVALID, "graph_segment_sum: bad sizes");
    if (g->num_graphs == 0 || F == 0) return GVQA_OK;
    GVQA_REQUIRE((x || g->num_nodes == 0) && out, GVQA_E_INVALID, "graph_segment_sum: null tensor");
    const dim3 grid((unsigned)g->num_graphs, (unsigned)cdiv(F, 256));
    hipLaunchKernelGGL(k_graph_segment_sum, grid, dim3(256), 0, static_cast<hipStream_t>(stream), (int)F, g->graph_ptr, x, ld_x, out,
                       ld_out, mean);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}
}  // namespace gvqa

int gvqa_graph_segment_sum(const gvqa_graph* g, int64_t F, const float* x, int64_t ld_x, float* out, int64_t ld_out, void* stream) {
    return gvqa::graph_segment_reduce(g, F, x, ld_x, out, ld_out, 0, stream);
}

int gvqa_graph_segment_mean(const gvqa_graph* g, int64_t F, const float* x, int64_t ld_x, float* out, int64_t ld_out, void* stream) {
    return gvqa::graph_segment_reduce(g, F, x, ld_x, out, ld_out, 1, stream);
}

}  // extern "C"
