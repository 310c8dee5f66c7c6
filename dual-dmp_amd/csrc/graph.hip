// Graph handle: CSR of A + I with D^-1/2, built once (the reference recomputes gcn_norm on every
// GCNConv call, 24x per step, on a static mesh graph).  Host-side counting sort; one-off.
#include "ddmp_common.h"

#include <algorithm>
#include <new>
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdlib>

extern "C" int ddmp_abi_version(void) { return DDMP_ABI_VERSION; }

extern "C" const char* ddmp_status_string(int status) {
    switch (status) {
        case DDMP_OK: return "ok";
        case DDMP_EINVAL: return "invalid argument";
        case DDMP_ERANGE: return "index out of range";
        case DDMP_ENOMEM: return "host allocation failed";
        case DDMP_EWORKSPACE: return "workspace too small";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

extern "C" int ddmp_csr_build_host(int64_t n, int64_t nnz, const int64_t* ei, int32_t* rowptr,
                                   int32_t* col, float* dinv, int64_t* nnz_out) {
    ARG_TRY(n > 0 && nnz >= 0 && rowptr && col && dinv && nnz_out);
    ARG_TRY(nnz == 0 || ei);
    ARG_TRY(n < (int64_t)INT32_MAX && nnz + n < (int64_t)INT32_MAX);
    const int64_t* src = ei;
    const int64_t* dst = ei + nnz;
    std::vector<int32_t> cnt;
    try { cnt.assign((size_t)n + 1, 0); } catch (const std::bad_alloc&) { return DDMP_ENOMEM; }
    int64_t kept = 0;
    for (int64_t e = 0; e < nnz; ++e) {
        const int64_t s = src[e], d = dst[e];
        if (s < 0 || s >= n || d < 0 || d >= n) return DDMP_ERANGE;
        if (s == d) continue;                         // add_remaining_self_loops drops explicit loops
        cnt[(size_t)d + 1]++;
        kept++;
    }
    if (*nnz_out < kept + n) return DDMP_EWORKSPACE;
    rowptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t deg = cnt[(size_t)i + 1] + 1;   // + the self loop
        rowptr[i + 1] = rowptr[i] + deg;
        dinv[i] = (float)(1.0 / std::sqrt((double)deg));
    }
    // fill: cursor per row
    for (int64_t i = 0; i < n; ++i) cnt[(size_t)i] = rowptr[i];
    for (int64_t e = 0; e < nnz; ++e) {
        const int64_t s = src[e], d = dst[e];
        if (s == d) continue;
        col[cnt[(size_t)d]++] = (int32_t)s;
    }
    for (int64_t i = 0; i < n; ++i) {
        col[cnt[(size_t)i]++] = (int32_t)i;
        std::sort(col + rowptr[i], col + rowptr[i + 1]);
    }
    *nnz_out = kept + n;
    return DDMP_OK;
}

extern "C" int ddmp_csr_bfs_order_host(int64_t n, const int32_t* rowptr, const int32_t* col,
                                       int32_t* order) {
    ARG_TRY(n > 0 && rowptr && col && order);
    std::vector<uint8_t> seen;
    try { seen.assign((size_t)n, 0); } catch (const std::bad_alloc&) { return DDMP_ENOMEM; }
    int64_t head = 0, tail = 0;
    for (int64_t seed = 0; seed < n; ++seed) {
        if (seen[(size_t)seed]) continue;
        seen[(size_t)seed] = 1;
        order[tail++] = (int32_t)seed;
        while (head < tail) {
            const int32_t u = order[head++];
            for (int32_t e = rowptr[u]; e < rowptr[u + 1]; ++e) {
                const int32_t v = col[e];
                if (v < 0 || v >= n) return DDMP_ERANGE;
                if (!seen[(size_t)v]) {
                    seen[(size_t)v] = 1;
                    order[tail++] = v;
                }
            }
        }
    }
    return DDMP_OK;
}

namespace {
struct RcbCtx {
    const double* p;
    int leaf;
};
void rcb_split(const RcbCtx& c, int32_t* idx, int64_t m) {
    while (m > c.leaf) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int64_t i = 0; i < m; ++i)
            for (int a = 0; a < 3; ++a) {
                const double v = c.p[3 * (int64_t)idx[i] + a];
                lo[a] = std::min(lo[a], v);
                hi[a] = std::max(hi[a], v);
            }
        int ax = 0;
        for (int a = 1; a < 3; ++a)
            if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
        const int64_t leaves = (m + c.leaf - 1) / c.leaf;
        const int64_t nl = (leaves / 2) * c.leaf;               // the left part: whole leaves
        const double* p = c.p;
        std::nth_element(idx, idx + nl, idx + m, [p, ax](int32_t u, int32_t v) {
            const double a = p[3 * (int64_t)u + ax], b = p[3 * (int64_t)v + ax];
            return a < b || (a == b && u < v);
        });
        rcb_split(c, idx, nl);                                  // (recursion depth <= log2(n / leaf))
        idx += nl;
        m -= nl;
    }
}
}  // namespace

extern "C" int ddmp_rcb_order_host(int64_t n, const double* xyz, int leaf, int32_t* order) {
    ARG_TRY(n > 0 && n < (int64_t)INT32_MAX && xyz && order && leaf > 0);
    for (int64_t i = 0; i < n; ++i) order[i] = (int32_t)i;
    for (int64_t i = 0; i < 3 * n; ++i)
        if (!(xyz[i] == xyz[i])) return DDMP_EINVAL;            // NaN coordinates have no order
    rcb_split(RcbCtx{xyz, leaf}, order, n);
    return DDMP_OK;
}

static int upload_graph(int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                        const float* dinv, ddmp_graph** out, int64_t row0 = 0) {
    ddmp_graph* g = new (std::nothrow) ddmp_graph();
    if (!g) return DDMP_ENOMEM;
    std::memset(g, 0, sizeof(*g));
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = rowptr[n_rows];
    int mx = 0;
    for (int64_t i = 0; i < n_rows; ++i) mx = std::max(mx, (int)(rowptr[i + 1] - rowptr[i]));
    g->max_row_nnz = mx;
    hipError_t e;
    if ((e = hipMalloc((void**)&g->rowptr, sizeof(int32_t) * (size_t)(n_rows + 1))) != hipSuccess) goto fail;
    if ((e = hipMalloc((void**)&g->col, sizeof(int32_t) * (size_t)std::max<int64_t>(g->nnz, 1))) != hipSuccess) goto fail;
    if ((e = hipMalloc((void**)&g->dinv, sizeof(float) * (size_t)n_cols)) != hipSuccess) goto fail;
    if ((e = hipMemcpy(g->rowptr, rowptr, sizeof(int32_t) * (size_t)(n_rows + 1), hipMemcpyHostToDevice)) != hipSuccess) goto fail;
    if (g->nnz > 0 &&
        (e = hipMemcpy(g->col, col, sizeof(int32_t) * (size_t)g->nnz, hipMemcpyHostToDevice)) != hipSuccess) goto fail;
    if ((e = hipMemcpy(g->dinv, dinv, sizeof(float) * (size_t)n_cols, hipMemcpyHostToDevice)) != hipSuccess) goto fail;
    g->dinv_r = g->dinv + row0;                                  // output row i is node row0 + i of the column numbering
    {   // the entries' weights dinv[col[e]] as the gather kernels stage them: no col -> dinv chain in a chunk's set-up
        std::vector<float> ew((size_t)std::max<int64_t>(g->nnz, 1), 0.f);
        for (int64_t e2 = 0; e2 < g->nnz; ++e2) ew[(size_t)e2] = dinv[col[e2]];
        if ((e = hipMalloc((void**)&g->ew, sizeof(float) * ew.size())) != hipSuccess) goto fail;
        if ((e = hipMemcpy(g->ew, ew.data(), sizeof(float) * ew.size(), hipMemcpyHostToDevice)) != hipSuccess) goto fail;
    }
    {   // patch tables (see ddmp_graph): skipped when the rows of a chunk fan out too far (unordered numbering)
        const int64_t n_chunks = (n_rows + ddmp::kChunkRows - 1) / ddmp::kChunkRows;
        // by default from 64k rows for graphs with at most 12 entries per row ON AVERAGE (mesh graphs: 4 and ~7; the LDS
        // tables of a chunk hold 16 per row); DDMP_SPMM_PATCH=1 every graph (A/B), =0 none.  Row lengths are NOT a condition
        // any more (round 5): long rows run their tail from LDS lists, oversized chunks go to the heavy list.
        const char* pe = getenv("DDMP_SPMM_PATCH");
        const int pm = pe ? atoi(pe) : 3;
        const int64_t min_rows = 65536;                          // (16384 measured no gain at 62,500 rows: profiles/r06_launch_bound_probes.txt)
        bool ok = g->nnz > 0 && (pm == 1 || (pm != 0 && g->nnz <= 12 * n_rows && n_rows >= min_rows));
        constexpr int kMaxE = ddmp::kChunkRows * 16, kMaxKd = 6;
        std::vector<int32_t> np_of, tmp;
        if (ok) {
            np_of.assign((size_t)n_chunks, 0);
            int64_t over = 0;                                    // chunks that cannot be taken even with the largest buffers
            for (int64_t c = 0; c < n_chunks; ++c) {
                const int64_t r0 = c * ddmp::kChunkRows, r1 = std::min<int64_t>(n_rows, r0 + ddmp::kChunkRows);
                const int64_t ne = rowptr[r1] - rowptr[r0];
                if (ne == 0 || ne > kMaxE) { np_of[(size_t)c] = ne == 0 ? 0 : INT32_MAX; over += ne != 0; continue; }
                tmp.assign(col + rowptr[r0], col + rowptr[r1]);
                std::sort(tmp.begin(), tmp.end());
                const int np = (int)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
                np_of[(size_t)c] = np;
                over += np > 32 * kMaxKd;
                if (c == 1023 && over > 512) { over = n_chunks; break; }     // an unordered numbering: stop counting
            }
            ok = over * 50 <= n_chunks;                          // more than 2 % oversized: not a graph for this kernel
        }
        if (ok) {
            int kd = 3;                                          // the smallest buffers that leave at most 1 % of the chunks heavy
            for (; kd < kMaxKd; ++kd) {
                int64_t heavy = 0;
                for (int64_t c = 0; c < n_chunks; ++c) heavy += np_of[(size_t)c] > 32 * kd;
                if (heavy * 100 <= n_chunks) break;
            }
            std::vector<int32_t> pl_ptr((size_t)n_chunks + 1, 0), pl_col, heavy, split((size_t)2 * n_chunks, 0);
            int n_split = 0;
            std::vector<uint16_t> lcol((size_t)std::max<int64_t>(g->nnz, 1), 0);
            int max_patch = 0;
            for (int64_t c = 0; c < n_chunks; ++c) {
                const int np = np_of[(size_t)c];
                const int64_t c0 = c * ddmp::kChunkRows, c1 = std::min<int64_t>(n_rows, c0 + ddmp::kChunkRows);
                int parts = 0;                                   // too large a patch: do its 2 halves (32 rows) or 4 quarters (16 rows) fit?
                if (np > 32 * kd && np != INT32_MAX) {
                    for (int nh = 2; nh <= 4 && !parts; nh *= 2) {
                        const int64_t rp = ddmp::kChunkRows / nh;
                        bool fit = c1 - c0 > rp * (nh - 1);      // (every part has rows)
                        for (int h = 0; h < nh && fit; ++h) {
                            const int64_t a0 = c0 + rp * h, a1 = std::min<int64_t>(c1, a0 + rp);
                            tmp.assign(col + rowptr[a0], col + rowptr[a1]);
                            std::sort(tmp.begin(), tmp.end());
                            const int64_t u = std::unique(tmp.begin(), tmp.end()) - tmp.begin();
                            fit = u > 0 && u <= 32 * kd;         // (a part without entries has no patch row to point at)
                        }
                        if (fit) parts = nh;
                    }
                }
                if (parts) {
                    const int64_t rp = ddmp::kChunkRows / parts;
                    g->max_chunk_nnz = std::max(g->max_chunk_nnz, (int)(rowptr[c1] - rowptr[c0]));
                    split[(size_t)2 * c] = (int32_t)split.size();            // this chunk's (patch start, record slot) pairs, parts 1 ..
                    split[(size_t)2 * c + 1] = parts;
                    for (int h = 0; h < parts; ++h) {
                        const int64_t a0 = c0 + rp * h, a1 = std::min<int64_t>(c1, a0 + rp);
                        tmp.assign(col + rowptr[a0], col + rowptr[a1]);
                        std::sort(tmp.begin(), tmp.end());
                        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
                        max_patch = std::max(max_patch, (int)tmp.size());
                        for (int64_t e2 = rowptr[a0]; e2 < rowptr[a1]; ++e2)
                            lcol[(size_t)e2] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), col[e2]) - tmp.begin());
                        if (h > 0) {
                            split.push_back((int32_t)pl_col.size());
                            split.push_back((int32_t)(n_chunks + n_split++));
                        }
                        pl_col.insert(pl_col.end(), tmp.begin(), tmp.end());
                    }
                } else if (np == 0 || np > 32 * kd) {
                    heavy.push_back((int32_t)c);
                } else {
                    const int64_t r0 = c * ddmp::kChunkRows, r1 = std::min<int64_t>(n_rows, r0 + ddmp::kChunkRows);
                    tmp.assign(col + rowptr[r0], col + rowptr[r1]);
                    std::sort(tmp.begin(), tmp.end());
                    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
                    max_patch = std::max(max_patch, (int)tmp.size());
                    g->max_chunk_nnz = std::max(g->max_chunk_nnz, (int)(rowptr[r1] - rowptr[r0]));
                    for (int64_t e2 = rowptr[r0]; e2 < rowptr[r1]; ++e2)
                        lcol[(size_t)e2] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), col[e2]) - tmp.begin());
                    pl_col.insert(pl_col.end(), tmp.begin(), tmp.end());
                }
                pl_ptr[(size_t)c + 1] = (int32_t)pl_col.size();
            }
            if (max_patch > 0) {
                if ((e = hipMalloc((void**)&g->pl_ptr, sizeof(int32_t) * pl_ptr.size())) != hipSuccess) goto fail;
                if ((e = hipMalloc((void**)&g->pl_col, sizeof(int32_t) * std::max<size_t>(pl_col.size(), 1))) != hipSuccess) goto fail;
                if ((e = hipMalloc((void**)&g->lcol, sizeof(uint16_t) * lcol.size())) != hipSuccess) goto fail;
                if ((e = hipMemcpy(g->pl_ptr, pl_ptr.data(), sizeof(int32_t) * pl_ptr.size(), hipMemcpyHostToDevice)) != hipSuccess) goto fail;
                if ((e = hipMemcpy(g->pl_col, pl_col.data(), sizeof(int32_t) * pl_col.size(), hipMemcpyHostToDevice)) != hipSuccess) goto fail;
                if ((e = hipMemcpy(g->lcol, lcol.data(), sizeof(uint16_t) * lcol.size(), hipMemcpyHostToDevice)) != hipSuccess) goto fail;
                if (n_split > 0) {
                    if ((e = hipMalloc((void**)&g->pl_split, sizeof(int32_t) * split.size())) != hipSuccess) goto fail;
                    if ((e = hipMemcpy(g->pl_split, split.data(), sizeof(int32_t) * split.size(), hipMemcpyHostToDevice)) != hipSuccess) goto fail;
                    g->n_split = n_split;
                }
                if (!heavy.empty()) {
                    if ((e = hipMalloc((void**)&g->heavy, sizeof(int32_t) * heavy.size())) != hipSuccess) goto fail;
                    if ((e = hipMemcpy(g->heavy, heavy.data(), sizeof(int32_t) * heavy.size(), hipMemcpyHostToDevice)) != hipSuccess) goto fail;
                }
                g->n_heavy = (int)heavy.size();
                g->max_patch = max_patch;
                g->patch_kd = kd;
            }
        }
    }
    *out = g;
    return DDMP_OK;
fail:
    ddmp_graph_destroy(g);
    return (int)e;
}

extern "C" int ddmp_graph_create_csr_host(int64_t n_rows, int64_t n_cols, const int32_t* rowptr,
                                          const int32_t* col, const float* dinv, ddmp_graph** out) {
    ARG_TRY(out && n_rows > 0 && n_cols >= n_rows && rowptr && col && dinv);
    ARG_TRY(rowptr[0] == 0);
    for (int64_t i = 0; i < n_rows; ++i) ARG_TRY(rowptr[i + 1] >= rowptr[i]);
    for (int64_t e = 0; e < rowptr[n_rows]; ++e)
        if (col[e] < 0 || col[e] >= n_cols) return DDMP_ERANGE;
    return upload_graph(n_rows, n_cols, rowptr, col, dinv, out);
}

// Rows [row0, row1) of a local CSR as a graph of their own (round 6: the interior / boundary halves of a partitioned graph,
// so that the halo exchange of a layer travels while the interior rows are aggregated -- dual-dmp_amd/dist.py): output row i
// is node row0 + i, the columns keep the numbering of the whole local graph (X is the same tensor, Y starts at row row0).
extern "C" int ddmp_graph_create_csr_rows_host(int64_t n_rows_all, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                                               const float* dinv, int64_t row0, int64_t row1, ddmp_graph** out) {
    ARG_TRY(out && n_rows_all > 0 && n_cols >= n_rows_all && rowptr && col && dinv && row0 >= 0 && row1 > row0 && row1 <= n_rows_all);
    ARG_TRY(rowptr[0] == 0);
    for (int64_t i = 0; i < n_rows_all; ++i) ARG_TRY(rowptr[i + 1] >= rowptr[i]);
    for (int64_t e = rowptr[row0]; e < rowptr[row1]; ++e)
        if (col[e] < 0 || col[e] >= n_cols) return DDMP_ERANGE;
    std::vector<int32_t> rp;
    try {
        rp.resize((size_t)(row1 - row0) + 1);
    } catch (const std::bad_alloc&) {
        return DDMP_ENOMEM;
    }
    for (int64_t i = row0; i <= row1; ++i) rp[(size_t)(i - row0)] = rowptr[i] - rowptr[row0];
    return upload_graph(row1 - row0, n_cols, rp.data(), col + rowptr[row0], dinv, out, row0);
}

extern "C" int ddmp_graph_create(int64_t n, int64_t nnz, const int64_t* edge_index, int on_device,
                                 ddmp_graph** out) {
    ARG_TRY(out && n > 0 && nnz >= 0 && (nnz == 0 || edge_index));
    std::vector<int64_t> host_ei;
    std::vector<int32_t> rowptr, col;
    std::vector<float> dinv;
    try {
        rowptr.resize((size_t)n + 1);
        col.resize((size_t)(nnz + n));
        dinv.resize((size_t)n);
        if (on_device && nnz > 0) host_ei.resize((size_t)(2 * nnz));
    } catch (const std::bad_alloc&) {
        return DDMP_ENOMEM;
    }
    const int64_t* ei = edge_index;
    if (on_device && nnz > 0) {
        HIP_TRY(hipMemcpy(host_ei.data(), edge_index, sizeof(int64_t) * (size_t)(2 * nnz), hipMemcpyDeviceToHost));
        ei = host_ei.data();
    }
    int64_t used = nnz + n;
    int st = ddmp_csr_build_host(n, nnz, ei, rowptr.data(), col.data(), dinv.data(), &used);
    if (st != DDMP_OK) return st;
    return upload_graph(n, n, rowptr.data(), col.data(), dinv.data(), out);
}

extern "C" int ddmp_graph_destroy(ddmp_graph* g) {
    if (!g) return DDMP_OK;
    if (g->rowptr) (void)hipFree(g->rowptr);
    if (g->col) (void)hipFree(g->col);
    if (g->dinv) (void)hipFree(g->dinv);
    if (g->pl_ptr) (void)hipFree(g->pl_ptr);
    if (g->pl_col) (void)hipFree(g->pl_col);
    if (g->lcol) (void)hipFree(g->lcol);
    if (g->ew) (void)hipFree(g->ew);
    if (g->pl_split) (void)hipFree(g->pl_split);
    if (g->heavy) (void)hipFree(g->heavy);
    delete g;
    return DDMP_OK;
}

extern "C" int ddmp_graph_info(const ddmp_graph* g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz,
                               int* max_row_nnz) {
    ARG_TRY(g);
    if (n_rows) *n_rows = g->n_rows;
    if (n_cols) *n_cols = g->n_cols;
    if (nnz) *nnz = g->nnz;
    if (max_row_nnz) *max_row_nnz = g->max_row_nnz;
    return DDMP_OK;
}

extern "C" int ddmp_graph_patch_info(const ddmp_graph* g, int* patch_kd, int* n_heavy, int* n_split) {
    ARG_TRY(g);
    if (patch_kd) *patch_kd = g->max_patch > 0 ? g->patch_kd : 0;
    if (n_heavy) *n_heavy = g->n_heavy;
    if (n_split) *n_split = g->n_split;
    return DDMP_OK;
}

extern "C" int ddmp_graph_tables(const ddmp_graph* g, const int32_t** rowptr, const int32_t** col,
                                 const float** dinv) {
    ARG_TRY(g);
    if (rowptr) *rowptr = g->rowptr;
    if (col) *col = g->col;
    if (dinv) *dinv = g->dinv;
    return DDMP_OK;
}
