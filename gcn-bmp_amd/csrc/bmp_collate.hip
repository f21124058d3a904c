// The reference's per-iteration collate (SerialIterator + converter=concat_mols, train_ddi_modify.py:280,295-296) for a
// drug store that lives in HBM.  A batch of drug pairs is index work only: the store holds every molecule's atom ids
// and its local CSR / transposed CSR (sorted by destination, source, bond type inside the molecule); a packed batch
// (bmp/packed.py) is those per-molecule arrays shifted by the molecule's first packed row, because molecules own
// disjoint, contiguous row ranges and the batch CSR order (row, source row, bond type) is the per-molecule order of
// the molecules taken in row order.  So no sort is needed per batch:
//   * bmp_collate_plan (HOST, no device work): the size arithmetic -- per-side zero-padding width A, first-fit-decreasing
//     placement of the instances into 128-row tiles (the same placement as bmp.packed._bin_pack, which the tests pin it
//     against), edge base of every instance, dead rows behind the last instance of a tile, and the co-attention's
//     per-pair metadata (C-block offsets, size-class orders).  O(I log I) on a few thousand small integers.
//   * bmp_collate_emit (DEVICE): one wavefront per molecule instance writes atom_id, row_w, row_mol, csr_ptr/col/val and
//     csrT_ptr/col/val.  Integer / byte work, HBM bound (~12 bytes per row and ~16 per bond written, the store reads hit L2).
// The result is bit-identical to bmp.packed.pack_from_store (tests/test_gpu_collate.py).
#include <algorithm>
#include <vector>
#include "bmp_common.h"

namespace {

// leftmost bin with capacity >= s: max segment tree over bin capacities
struct FirstFit {
    int n, base;
    std::vector<int> t;
    FirstFit(int n_, int cap) : n(n_) {
        base = 1;
        while (base < n) base <<= 1;
        t.assign(2 * base, 0);
        for (int i = 0; i < n; ++i) t[base + i] = cap;
        for (int i = base - 1; i >= 1; --i) t[i] = std::max(t[2 * i], t[2 * i + 1]);
    }
    int take(int s, int* off_before, int R) {
        int i = 1;
        while (i < base) i = 2 * i + (t[2 * i] < s);      // (arithmetic, not a branch: the direction is unpredictable)
        const int b = i - base;
        *off_before = R - t[i];
        t[i] -= s;
        // (upwards only while a maximum changes: the full walk is a chain of ten dependent store-to-load steps per item,
        //  and most placements leave the parent's maximum -- a fresh bin beside it -- as it was)
        for (i >>= 1; i >= 1; i >>= 1) {
            const int m = std::max(t[2 * i], t[2 * i + 1]);
            if (m == t[i]) break;
            t[i] = m;
        }
        return b;
    }
    int cap(int b) const { return t[base + b]; }
};

}  // namespace

// Layout of the plan table `tab` (int32, 6 * I entries): row0[I] | nrows[I] | mid[I] | ebase[I] | padw[I] | ndead[I].
//   row0  first packed row of the instance          nrows real atoms + 1 (the virtual pad row)
//   mid   molecule of the store                     ebase first CSR entry of the instance
//   padw  multiplicity of the virtual pad row = A[side] - n (concat_mols zero-padding)
//   ndead dead rows that follow the instance (non-zero only for the last instance of a tile)
// side_ptr [n_sides + 1]: instance ranges of the batch sides in `mids`; pad_to [n_sides] or NULL (pad to the side's max).
// side_tiles [n_sides + 1] (out): tile boundaries of the sides.  totals (out): n_tiles, n_edges, n_real_atoms,
// max rows of an instance.
extern "C" int bmp_collate_plan(const int* st_nrows, const int* st_nedges, int n_store, const int* mids, const int* side_ptr,
                                int n_sides, const int* pad_to, int R, int* tab, int* side_tiles, long long* totals) {
    BMP_REQUIRE(st_nrows && st_nedges && mids && side_ptr && tab && side_tiles && totals && n_sides >= 1 && R >= 1);
    const int I = side_ptr[n_sides];
    BMP_REQUIRE(I >= 1 && side_ptr[0] == 0);
    int* row0 = tab; int* nrows = tab + I; int* mid = tab + 2 * (size_t)I; int* ebase = tab + 3 * (size_t)I;
    int* padw = tab + 4 * (size_t)I; int* ndead = tab + 5 * (size_t)I;
    long long n_real = 0, n_edges = 0;
    int max_rows = 0;
    for (int i = 0; i < I; ++i) {
        BMP_REQUIRE(mids[i] >= 0 && mids[i] < n_store);
        mid[i] = mids[i];
        nrows[i] = st_nrows[mids[i]];
        BMP_REQUIRE(nrows[i] >= 1);
        n_real += nrows[i] - 1;
        n_edges += st_nedges[mids[i]];
        max_rows = std::max(max_rows, nrows[i]);
        ndead[i] = 0;
    }
    int tile0 = 0;
    side_tiles[0] = 0;
    // An instance of more than R rows (a molecule of more than R - 1 atoms: the reference's preprocessor has no size limit,
    // train_ddi_modify.py:256) takes ceil(rows / R) whole consecutive tiles of its own at the head of its side, in the stable
    // decreasing-size order; the rest of its last tile stays dead.  Same placement as bmp.packed._bin_pack.
    std::vector<int> order, cnt(R + 2), bigs, placed;
    placed.reserve(I);
    for (int s = 0; s < n_sides; ++s) {
        const int lo = side_ptr[s], hi = side_ptr[s + 1], n = hi - lo;
        BMP_REQUIRE(n >= 1);
        int A = 0;
        for (int i = lo; i < hi; ++i) A = std::max(A, nrows[i] - 1);
        if (pad_to) { BMP_REQUIRE(pad_to[s] >= A); A = pad_to[s]; }
        for (int i = lo; i < hi; ++i) padw[i] = A - (nrows[i] - 1);
        // oversized instances first: stable by decreasing size
        bigs.clear();
        for (int i = lo; i < hi; ++i) if (nrows[i] > R) bigs.push_back(i);
        std::stable_sort(bigs.begin(), bigs.end(), [&](int a, int b) { return nrows[a] > nrows[b]; });
        for (int it : bigs) {
            const int k = (nrows[it] + R - 1) / R;
            row0[it] = tile0 * R;
            ndead[it] = k * R - nrows[it];
            tile0 += k;
            placed.push_back(it);
        }
        const int n_small = n - (int)bigs.size();
        // stable order by decreasing size (counting sort; == numpy argsort(-sizes, kind="stable"))
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int i = lo; i < hi; ++i) if (nrows[i] <= R) ++cnt[R - nrows[i] + 1];
        for (int k = 1; k <= R + 1; ++k) cnt[k] += cnt[k - 1];
        order.assign(n_small, 0);
        for (int i = lo; i < hi; ++i) if (nrows[i] <= R) order[cnt[R - nrows[i]]++] = i;
        int nb = 0;
        if (n_small > 0) {
            FirstFit ff(n_small, R);
            std::vector<int> last(n_small, -1);
            for (int q = 0; q < n_small; ++q) {
                const int it = order[q];
                int off;
                const int b = ff.take(nrows[it], &off, R);
                row0[it] = (tile0 + b) * R + off;
                placed.push_back(it);
                last[b] = it;                   // offsets grow with every placement: the latest item is the tile's last
                nb = std::max(nb, b + 1);
            }
            for (int b = 0; b < nb; ++b) ndead[last[b]] = ff.cap(b);
        }
        tile0 += nb;
        side_tiles[s + 1] = tile0;
    }
    // edge bases: instances in packed row order.  `placed` lists them in placement order, where a tile's offsets only grow:
    // a stable counting sort of that list by tile IS the row order (no comparison sort: 0.1 ms of a 0.27 ms call before)
    std::vector<int> by_row(I), tcnt(tile0 + 1, 0);
    for (int i = 0; i < I; ++i) ++tcnt[row0[i] / R + 1];
    for (int t = 0; t < tile0; ++t) tcnt[t + 1] += tcnt[t];
    for (int q = 0; q < I; ++q) by_row[tcnt[row0[placed[q]] / R]++] = placed[q];
    long long e = 0;
    for (int q = 0; q < I; ++q) {
        ebase[by_row[q]] = (int)e;
        e += st_nedges[mid[by_row[q]]];
    }
    BMP_REQUIRE(e == n_edges && e < (1ll << 31) && (long long)tile0 * R < (1ll << 29));
    totals[0] = tile0; totals[1] = n_edges; totals[2] = n_real; totals[3] = max_rows;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The ENCODER LAYOUT of a batch (csrc/bmp_enc.hip): the real atoms of every encoded molecule, one pad row per tile, tiles of
// 1..4 live 32-row blocks, dense rows.  HOST function, size arithmetic only.
//
// Encoded molecules: every instance of `mids` (dedup == 0), or the DISTINCT molecules in ascending store order
// (dedup != 0: SURVEY.md 8(d) caveat).  Tile heights: the batch needs about sum(n) / 32 block-rounds of one CU; with
// b = that number per CU of the n_cu CUs (b <= 7), every CU is meant to get the heights DEC[b] (7 -> 4 + 3, 6 -> 3 + 3,
// 5 -> 3 + 2, else b itself): the bin list is n_cu bins of every height of DEC[b], tall first, followed by spare bins (as
// tall as their first molecule needs; 128 rows when b = 8, the uniform fallback); molecules go first-fit in stable
// decreasing-size order (a bin of height h holds 32 h - 1 atoms: one row stays for the pad row).  Tiles = the non-empty spare bins (they took what fitted nowhere else: the largest molecules), then the
// non-empty listed bins, in that order -- so the tallest tiles are dispatched first and a CU that finishes a short tile
// picks up the next one; a tile's height is what its molecules need, ceil((atoms + 1) / 32).  One-block dummy tiles pad the
// row count to a multiple of 128 (the row-wise kernels' granularity).  Mirrored by bmp.enclayout.plan_enc_numpy (tests).
//
// Outputs (all int32; U <= I encoded molecules, T <= U + 3 tiles):
//   tab [6 * I]   as bmp_collate_plan for the U encoded molecules (row0 | n | mid | ebase | 1 | ndead), stride U
//   tile_last [I] tile of a molecule that is the last of its tile, else -1 (bmp_collate_emit marks the pad row with it)
//   uid [I], uptr [I + 1], uinst [I]   encoded molecule of every instance; instances per encoded molecule, ascending
//   enc_pad [I]   pad row of every encoded molecule's tile
//   tptr [I + 4], tmols [I]            encoded molecules per tile, ascending
//   mt_row0 [I + 4], mt_nblk [I + 4]   the tile table
//   totals [6]: U, T, N_enc, n_edges, n_real (rows of encoded molecules), budget b (8: uniform 128-row bins)
// Returns -2000 if a molecule has more than R - 1 atoms (the caller keeps the per-instance layout for such a batch).
extern "C" int bmp_collate_plan_enc(const int* st_nrows, const int* st_nedges, int n_store, const int* mids, int I, int dedup,
                                    int n_cu, int R, int* tab, int* tile_last, int* uid, int* uptr, int* uinst, int* enc_pad,
                                    int* tptr, int* tmols, int* mt_row0, int* mt_nblk, long long* totals) {
    BMP_REQUIRE(st_nrows && st_nedges && mids && tab && tile_last && uid && uptr && uinst && enc_pad && tptr && tmols && mt_row0 &&
                mt_nblk && totals && I >= 1 && n_cu >= 1 && R == 128);
    // ---- encoded molecules ----
    std::vector<int> umid;
    if (dedup) {
        std::vector<int> seen(n_store, -1);
        for (int i = 0; i < I; ++i) { BMP_REQUIRE(mids[i] >= 0 && mids[i] < n_store); seen[mids[i]] = 0; }
        for (int m = 0; m < n_store; ++m) if (seen[m] == 0) { seen[m] = (int)umid.size(); umid.push_back(m); }
        for (int i = 0; i < I; ++i) uid[i] = seen[mids[i]];
    } else {
        umid.assign(mids, mids + I);
        for (int i = 0; i < I; ++i) { BMP_REQUIRE(mids[i] >= 0 && mids[i] < n_store); uid[i] = i; }
    }
    const int U = (int)umid.size();
    int* row0 = tab; int* nn = tab + U; int* mid = tab + 2 * (size_t)U; int* ebase = tab + 3 * (size_t)U;
    int* padw = tab + 4 * (size_t)U; int* ndead = tab + 5 * (size_t)U;
    long long tot = 0, n_edges = 0;
    for (int u = 0; u < U; ++u) {
        mid[u] = umid[u];
        nn[u] = st_nrows[umid[u]] - 1;
        if (nn[u] + 1 > R) return -2000;
        BMP_REQUIRE(nn[u] >= 1);
        tot += nn[u]; n_edges += st_nedges[umid[u]];
        padw[u] = 1; ndead[u] = 0; tile_last[u] = -1;
    }
    for (int i = U; i < I; ++i) tile_last[i] = -1;
    {   // instances per encoded molecule (counting sort, stable)
        for (int u = 0; u <= U; ++u) uptr[u] = 0;
        for (int i = 0; i < I; ++i) ++uptr[uid[i] + 1];
        for (int u = 0; u < U; ++u) uptr[u + 1] += uptr[u];
        std::vector<int> at(uptr, uptr + U);
        for (int i = 0; i < I; ++i) uinst[at[uid[i]]++] = i;
    }
    // ---- tile heights ----
    static const int DEC[8][2] = {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {4, 0}, {3, 2}, {3, 3}, {4, 3}};
    int b = 8;
    for (int k = 1; k <= 7; ++k) {
        long long cap = 0;
        for (int j = 0; j < 2; ++j) if (DEC[k][j]) cap += (long long)n_cu * (32 * DEC[k][j] - 1);
        if (100 * tot <= 99 * cap) { b = k; break; }
    }
    std::vector<int> caps;
    if (b < 8)
        for (int j = 0; j < 2; ++j) if (DEC[b][j]) for (int q = 0; q < n_cu; ++q) caps.push_back(32 * DEC[b][j] - 1);
    const int n_list = (int)caps.size();
    for (int q = 0; q < U; ++q) caps.push_back(R - 1);              // spare bins
    const int NB = (int)caps.size();
    // leftmost bin with room: max segment tree over the remaining capacities
    int base = 1;
    while (base < NB) base <<= 1;
    std::vector<int> tr(2 * base, -1);
    for (int q = 0; q < NB; ++q) tr[base + q] = caps[q];
    for (int q = base - 1; q >= 1; --q) tr[q] = std::max(tr[2 * q], tr[2 * q + 1]);
    std::vector<int> order(U), cnt(R + 2, 0), bin_of(U), off_in(U), used(NB, 0);
    for (int u = 0; u < U; ++u) ++cnt[R - nn[u] + 1];
    for (int k = 1; k <= R + 1; ++k) cnt[k] += cnt[k - 1];
    for (int u = 0; u < U; ++u) order[cnt[R - nn[u]]++] = u;         // stable, decreasing size
    for (int q = 0; q < U; ++q) {
        const int u = order[q], sz = nn[u];
        int i = 1;
        BMP_REQUIRE(tr[1] >= sz);
        while (i < base) i = 2 * i + (tr[2 * i] < sz);
        const int bn = i - base;
        bin_of[u] = bn; off_in[u] = used[bn];
        // a spare bin opened beside the listed ones is only as tall as its first molecule needs (the largest molecules of a
        // small batch get tiles of their own height instead of filling a few 128-row tiles)
        if (b < 8 && bn >= n_list && used[bn] == 0) tr[i] = 32 * ((sz + 1 + 31) / 32) - 1;
        used[bn] += sz; tr[i] -= sz;
        for (i >>= 1; i >= 1; i >>= 1) {
            const int m = std::max(tr[2 * i], tr[2 * i + 1]);
            if (m == tr[i]) break;
            tr[i] = m;
        }
    }
    // ---- tiles: non-empty spare bins first, then the non-empty listed bins ----
    std::vector<int> tile_of_bin(NB, -1);
    int T = 0, rows = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int lo = pass == 0 ? n_list : 0, hi = pass == 0 ? NB : n_list;
        for (int q = lo; q < hi; ++q) {
            if (used[q] == 0) continue;
            tile_of_bin[q] = T;
            mt_row0[T] = rows;
            mt_nblk[T] = (used[q] + 1 + 31) / 32;
            rows += 32 * mt_nblk[T];
            ++T;
        }
    }
    const int T_real = T;
    while (rows % R) { mt_row0[T] = rows; mt_nblk[T] = 1; rows += 32; ++T; }     // dummy tiles (no molecule)
    // molecules of every tile, ascending; rows; pad rows; dead rows
    for (int t = 0; t <= T; ++t) tptr[t] = 0;
    for (int u = 0; u < U; ++u) ++tptr[tile_of_bin[bin_of[u]] + 1];
    for (int t = 0; t < T; ++t) tptr[t + 1] += tptr[t];
    {
        std::vector<int> at(tptr, tptr + T);
        for (int u = 0; u < U; ++u) tmols[at[tile_of_bin[bin_of[u]]]++] = u;
    }
    std::vector<int> last_of_tile(T, -1), last_off(T, -1);
    for (int u = 0; u < U; ++u) {
        const int t = tile_of_bin[bin_of[u]];
        row0[u] = mt_row0[t] + off_in[u];
        enc_pad[u] = mt_row0[t] + used[bin_of[u]];
        if (off_in[u] > last_off[t]) { last_off[t] = off_in[u]; last_of_tile[t] = u; }
    }
    for (int t = 0; t < T_real; ++t) {
        const int u = last_of_tile[t];
        tile_last[u] = t;
        ndead[u] = mt_row0[t] + 32 * mt_nblk[t] - (row0[u] + nn[u]);
    }
    if (T > T_real) ndead[last_of_tile[T_real - 1]] += 32 * (T - T_real);      // the dummy tiles' rows: zero-filled by that molecule
    // edge bases: encoded molecules in row order
    // (offsets inside a bin grow in placement order, tiles own disjoint row ranges: the placement order, counting-sorted by
    //  tile, is the row order)
    std::vector<int> by_row(U);
    {
        std::vector<int> at(tptr, tptr + T);
        for (int q = 0; q < U; ++q) { const int u = order[q]; by_row[at[tile_of_bin[bin_of[u]]]++] = u; }
    }
    long long e = 0;
    for (int q = 0; q < U; ++q) { ebase[by_row[q]] = (int)e; e += st_nedges[mid[by_row[q]]]; }
    BMP_REQUIRE(e == n_edges && e < (1ll << 31));
    totals[0] = U; totals[1] = T; totals[2] = rows; totals[3] = n_edges; totals[4] = tot; totals[5] = b;
    return 0;
}

// Co-attention metadata of a two-sided batch of B pairs (instances [0, B) are side 1, [B, 2B) side 2), from the plan
// table: what bmp/coattention.py:pair_rows computes with numpy.  meta (8 * B int32 entries, 8-byte aligned):
//   coff[B] as int64 (= 2B int32) | r1[B] | n1[B] | r2[B] (relative to side 2's first row) | n2[B] |
//   order[B] (size classes ascending, stable) | order_f[B] (size classes descending, stable).
// counts [6]: pairs per size class ceil(max(n1, n2) / 32) - 1 for classes 0..3; counts[4]: pairs with a molecule of more than
// 128 rows (the pair kernels' fifth class, which works out of global memory); counts[5]: the largest row count among those
// pairs (0 without any).  ctotal: floats of all C blocks (C [n2 x n1] and the softmax statistics of its rows and columns,
// bmp_coattn_nie_fwd).
extern "C" int bmp_collate_pair_meta(const int* tab, int I, int B, int side1_tiles, int R, int* meta, int* counts,
                                     long long* ctotal) {
    BMP_REQUIRE(tab && meta && counts && ctotal && B >= 1 && I == 2 * B);
    const int* row0 = tab; const int* nrows = tab + I;
    long long* coff = reinterpret_cast<long long*>(meta);
    int* r1 = meta + 2 * (size_t)B; int* n1 = r1 + B; int* r2 = n1 + B; int* n2 = r2 + B; int* ord = n2 + B; int* ordf = ord + B;
    long long c = 0;
    int cnt[5] = {0, 0, 0, 0, 0};
    int maxbig = 0;
    std::vector<int> cls(B);
    for (int p = 0; p < B; ++p) {
        const int a = nrows[p], b = nrows[B + p];
        coff[p] = c;
        c += (long long)a * b + 2ll * (a + b);
        r1[p] = row0[p]; n1[p] = a;
        r2[p] = row0[B + p] - side1_tiles * R; n2[p] = b;
        int k = (std::max(a, b) + 31) / 32 - 1;
        BMP_REQUIRE(k >= 0);
        if (k > 3) { k = 4; maxbig = std::max(maxbig, std::max(a, b)); }
        cls[p] = k; ++cnt[k];
    }
    int start[5], startf[5];
    start[0] = 0;
    for (int k = 1; k < 5; ++k) start[k] = start[k - 1] + cnt[k - 1];
    startf[4] = 0;
    for (int k = 3; k >= 0; --k) startf[k] = startf[k + 1] + cnt[k + 1];
    for (int p = 0; p < B; ++p) { ord[start[cls[p]]++] = p; ordf[startf[cls[p]]++] = p; }
    for (int k = 0; k < 5; ++k) counts[k] = cnt[k];
    counts[5] = maxbig;
    *ctotal = c;
    return 0;
}

// One wavefront per molecule instance (4 per workgroup).  st_rowoff [M + 1] / st_eoff [M + 1]: row / entry ranges of the
// store's molecules; st_atom: atom ids (0 at the virtual pad row); st_rend / st_rendT: per store row, END of the row's
// entries inside the molecule (local); st_col / st_colT: local_row << 2 | bond type.
__global__ __launch_bounds__(256) void k_collate_emit(const int* __restrict__ tab, int I, const int* __restrict__ st_rowoff,
                                                      const int* __restrict__ st_eoff, const int* __restrict__ st_atom,
                                                      const int* __restrict__ st_rend, const int* __restrict__ st_rendT,
                                                      const int* __restrict__ st_col, const int* __restrict__ st_colT,
                                                      int* __restrict__ atom_id, float* __restrict__ row_w, int* __restrict__ row_mol,
                                                      int* __restrict__ csr_ptr, int* __restrict__ csr_col, float* __restrict__ csr_val,
                                                      int* __restrict__ csrT_ptr, int* __restrict__ csrT_col,
                                                      float* __restrict__ csrT_val, const int* __restrict__ tile_last) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= I) return;
    const int row0 = tab[i], nrows = tab[I + i], mid = tab[2 * (size_t)I + i], ebase = tab[3 * (size_t)I + i];
    const int padw = tab[4 * (size_t)I + i], ndead = tab[5 * (size_t)I + i];
    const int ro = st_rowoff[mid], eo = st_eoff[mid], ne = st_eoff[mid + 1] - eo;
    for (int l = lane; l < nrows; l += 64) {
        const int r = row0 + l;
        atom_id[r] = st_atom[ro + l];
        row_w[r] = (l == nrows - 1) ? (float)padw : 1.0f;
        row_mol[r] = i;
        csr_ptr[r + 1] = ebase + st_rend[ro + l];
        csrT_ptr[r + 1] = ebase + st_rendT[ro + l];
    }
    for (int k = lane; k < ndead; k += 64) {
        const int r = row0 + nrows + k;
        atom_id[r] = 0;
        row_w[r] = 0.f;
        // encoder layout (bmp_collate_plan_enc): the first row behind a tile's last molecule is the tile's pad row
        row_mol[r] = (k == 0 && tile_last != nullptr && tile_last[i] >= 0) ? -2 - tile_last[i] : -1;
        csr_ptr[r + 1] = ebase + ne;
        csrT_ptr[r + 1] = ebase + ne;
    }
    const int shift = row0 << 2;
    for (int e = lane; e < ne; e += 64) {
        csr_col[ebase + e] = st_col[eo + e] + shift;
        csr_val[ebase + e] = 1.0f;
        csrT_col[ebase + e] = st_colT[eo + e] + shift;
        csrT_val[ebase + e] = 1.0f;
    }
    if (row0 == 0 && lane == 0) { csr_ptr[0] = 0; csrT_ptr[0] = 0; }
}

// tab: the plan table of bmp_collate_plan (or bmp_collate_plan_enc, then with its tile_last [I]; else NULL), on the device.  Output arrays as in bmp/packed.py:PackedMolBatch
// (N = n_tiles * R rows; csr_ptr / csrT_ptr hold N + 1 entries; col / val hold n_edges).
extern "C" int bmp_collate_emit(const int* tab, int I, const int* st_rowoff, const int* st_eoff, const int* st_atom,
                                const int* st_rend, const int* st_rendT, const int* st_col, const int* st_colT, int* atom_id,
                                float* row_w, int* row_mol, int* csr_ptr, int* csr_col, float* csr_val, int* csrT_ptr,
                                int* csrT_col, float* csrT_val, const int* tile_last, hipStream_t st) {
    BMP_REQUIRE(tab && st_rowoff && st_eoff && st_atom && st_rend && st_rendT && st_col && st_colT && atom_id && row_w &&
                row_mol && csr_ptr && csr_col && csr_val && csrT_ptr && csrT_col && csrT_val && I >= 1);
    hipLaunchKernelGGL(k_collate_emit, dim3((I + 3) / 4), dim3(256), 0, st, tab, I, st_rowoff, st_eoff, st_atom, st_rend,
                       st_rendT, st_col, st_colT, atom_id, row_w, row_mol, csr_ptr, csr_col, csr_val, csrT_ptr, csrT_col,
                       csrT_val, tile_last);
    BMP_LAUNCH_CHECK();
    return 0;
}

// rescale_adj (models/relgcn.py:20-28) on the packed CSR: every bond value is divided by the degree of its SOURCE atom
// (sum of adj over bond types and destination rows; 0 -> 1), as value * (1 / degree) -- the reference's rounding.  A source's
// bonds are one row of the transposed CSR, so its degree is that row's sum (a handful of entries): every thread recomputes
// the degree it needs, in the row's order, and no pass over a degree array is needed.
//   threads [0, E): entry t of the CSR;   threads [E, E + N): row (t - E) of the transposed CSR.
__global__ __launch_bounds__(256) void k_rescale_adj(const int* __restrict__ csr_col, const float* __restrict__ csr_val, int E,
                                                     const int* __restrict__ csrT_ptr, const float* __restrict__ csrT_val, int N,
                                                     float* __restrict__ out, float* __restrict__ outT) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < E) {
        const int src = csr_col[t] >> 2;
        float deg = 0.f;
        for (int k = csrT_ptr[src]; k < csrT_ptr[src + 1]; ++k) deg += csrT_val[k];
        out[t] = csr_val[t] * (1.0f / (deg != 0.f ? deg : 1.0f));
    } else if (t < E + N) {
        const int j = t - E;
        const int k0 = csrT_ptr[j], k1 = csrT_ptr[j + 1];
        float deg = 0.f;
        for (int k = k0; k < k1; ++k) deg += csrT_val[k];
        const float inv = 1.0f / (deg != 0.f ? deg : 1.0f);
        for (int k = k0; k < k1; ++k) outT[k] = csrT_val[k] * inv;
    }
}

extern "C" int bmp_rescale_adj(const int* csr_col, const float* csr_val, int E, const int* csrT_ptr, const float* csrT_val, int N,
                               float* csr_val_out, float* csrT_val_out, hipStream_t st) {
    BMP_REQUIRE(E >= 0 && N > 0 && csrT_ptr && (E == 0 || (csr_col && csr_val && csrT_val && csr_val_out && csrT_val_out)));
    if (E == 0) return 0;
    hipLaunchKernelGGL(k_rescale_adj, dim3((E + N + 255) / 256), dim3(256), 0, st, csr_col, csr_val, E, csrT_ptr, csrT_val, N,
                       csr_val_out, csrT_val_out);
    BMP_LAUNCH_CHECK();
    return 0;
}
