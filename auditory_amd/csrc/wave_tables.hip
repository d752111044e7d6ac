// Host side of the wave-autonomous kernels: the tables a plan uploads for them (one LDS-image blob per plan).
#include "capi_internal.h"

namespace audc {

// Tables of the wave-autonomous kernels (melspec_wave.hip) as one blob that is copied verbatim into LDS:
//   w4     per filter group one row of FLOAT32 weights (both compute types; x 1/4, exact): the group's filters (its
//          slots) one after the other, each as aligned 4-bin chunks (zero weights outside [lo, hi]), slot k padded to
//          slot_steps[k] chunks in every group
//   slots  per group and slot: the filter's first P chunk and its id (w64x16: compact rows, see below)
//   twa    pass twiddles W_N^(2 j k1), [k1 - 1][j]; tws: split twiddles W_N^k, k <= N/4 (compute type)
// A plan whose tables do not fit (16-bit indices, LDS) simply has no wave kernel.
int build_wave_tables(aud_plan* p, const int32_t* bin_pts, const double* mel_filters) {
    aud_ctx* c = p->ctx;
    const aud_plan_desc& d = p->d;
    const int N = d.win_samples, nf = d.mel.n_filters, dt = d.compute_dtype;
    const int kind = aud::melspec_wave_kind(N);  // N = 512 -> w16x16; N = 400 -> w20x10; N = 2048 -> w64x16
    aud::WaveGeometry g;
    if (!kind || !aud::melspec_wave_geometry(kind, N, &g)) return AUD_OK;
    const size_t tsz = dt == AUD_F64 ? 8 : 4;  // twiddles
    const size_t wsz = 4;                       // weights: float32
    const int G = g.n_groups, p_chunks = (N / 2 + 1 + 3) / 4;  // chunks of a padded power row (kHp / 4 of the kernel)
    if (nf >= 0xFFFF || nf > 8 * G) return AUD_OK;  // more than eight filters per group: no wave kernel
    // chunks per filter; a filter without taps still takes one (all-zero) step: its sum is 0 + LogOff
    std::vector<int> c0(nf), nc(nf), order(nf);
    for (int f = 0; f < nf; ++f) {
        const int lo = bin_pts[f], hi = bin_pts[f + 2];
        c0[f] = hi >= lo ? lo >> 2 : 0;
        nc[f] = hi >= lo ? (hi >> 2) - (lo >> 2) + 1 : 1;
        if (c0[f] + nc[f] > p_chunks) return AUD_OK;  // table reaches past the spectrum: the generic path reports it
        order[f] = f;
    }
    // widest filters first, dealt round-robin: slot k of every group then holds filters of nearly equal width
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nc[x] > nc[y]; });
    const int n_slots = std::max(1, (nf + G - 1) / G);
    aud::WaveArgs e{};
    int n_steps = 0;
    std::vector<int> slot_pos(n_slots);
    for (int k = 0; k < n_slots; ++k) {
        int mx = 1;
        for (int r = k * G; r < std::min(nf, (k + 1) * G); ++r) mx = std::max(mx, nc[order[r]]);
        if (mx > 255) return AUD_OK;
        e.slot_steps[k] = static_cast<unsigned char>(mx);
        slot_pos[k] = n_steps;
        n_steps += mx;
    }
    const bool compact = kind == 4;  // one filter group per lane: per-group rows of slot_steps chunks would be 2x the LDS
    // ds_read_b128 serves a wave in four groups of 16 lanes; lanes of one group that read 16-byte pieces with the same
    // index mod 16 conflict.  With one filter per lane and slot (w64x16) WHICH lane takes which filter of a slot is free:
    // deal the slot's filters so that the first P chunks of a lane group differ mod 16 wherever the table allows it (the
    // epilogue's P reads were 3-way on average with filters in width order)
    static const unsigned char kGroupOfLane[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0,
                                                   1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
    auto lane_group = [&](int gi) { return 2 * (gi >> 5) + kGroupOfLane[gi & 31]; };
    if (compact && G == 64) {
        for (int k = 0; k < n_slots; ++k) {
            const int lo_r = k * G, hi_r = std::min(nf, (k + 1) * G), ns = e.slot_steps[k];
            std::vector<int> fl(order.begin() + lo_r, order.begin() + hi_r);
            auto residue = [&](int f) { return std::min(c0[f], p_chunks - ns) & 15; };
            int count[16] = {0};
            for (int f : fl) ++count[residue(f)];
            // the most crowded residues choose their lane groups first
            std::stable_sort(fl.begin(), fl.end(), [&](int x, int y) { return count[residue(x)] > count[residue(y)]; });
            int have[4][16] = {{0}}, filled[4] = {0, 0, 0, 0};
            std::vector<int> lanes_of[4];
            for (int gi = 0; gi < G; ++gi) lanes_of[lane_group(gi)].push_back(gi);
            std::vector<int> at(G, -1);
            for (int f : fl) {
                int best = -1;
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (filled[g4] >= int(lanes_of[g4].size())) continue;
                    if (best < 0 || have[g4][residue(f)] < have[best][residue(f)] ||
                        (have[g4][residue(f)] == have[best][residue(f)] && filled[g4] < filled[best]))
                        best = g4;
                }
                at[lanes_of[best][filled[best]++]] = f;
                ++have[best][residue(f)];
            }
            // a short last slot leaves lanes empty: keep the filters on the lowest lanes of their groups, holes last
            int r = lo_r;
            std::vector<int> rest;
            for (int gi = 0; gi < G; ++gi)
                if (at[gi] >= 0) rest.push_back(at[gi]);
            if (hi_r - lo_r == G) {
                for (int gi = 0; gi < G; ++gi) order[r++] = at[gi];
            } else {
                for (int f : rest) order[r++] = f;
            }
        }
    }
    size_t w_stride = 0;
    std::vector<double> wrows;
    std::vector<uint32_t> slots;
    if (!compact) {
        // weight rows: row stride an odd number of 16-byte pieces, so that the groups' reads of one step spread over the banks
        w_stride = size_t(n_steps) * 4 * wsz;
        if ((w_stride / 16) % 2 == 0) w_stride += 16;
        wrows.assign(size_t(G) * (w_stride / wsz), 0.0);
        slots.assign(size_t(G) * n_slots, 0xFFFFu << 16);
    } else {
        // compact rows: every filter keeps only its own chunks (from the first P chunk its slot reads), one shared all-zero
        // chunk serves the steps past a filter's end; slot record = {first P chunk | filter id << 16, row start in 16-byte
        // pieces | own chunks << 16}
        slots.assign(size_t(G) * n_slots * 2, 0u);
        for (int gi = 0; gi < G; ++gi)
            for (int k = 0; k < n_slots; ++k) slots[(size_t(gi) * n_slots + k) * 2] = 0xFFFFu << 16;
        wrows.assign(4, 0.0);  // the zero chunk, at piece 0
    }
    size_t wpieces = 4 * wsz / 16;        // compact rows: 16-byte pieces laid down so far
    unsigned used[8][4] = {};             // [slot][16-lane read group]: piece residues mod 16 taken
    for (int r = 0; r < nf; ++r) {
        const int f = order[r], k = r / G, gi = r % G, ns = e.slot_steps[k];
        const int lo = bin_pts[f], hi = bin_pts[f + 2];
        const int pc0 = std::min(c0[f], p_chunks - ns);  // every step of the slot reads inside the row
        double* wr;
        if (!compact) {
            slots[size_t(gi) * n_slots + k] = uint32_t(pc0) | (uint32_t(f) << 16);
            wr = &wrows[size_t(gi) * (w_stride / wsz) + size_t(slot_pos[k]) * 4];
        } else {
            const int own = hi >= lo ? c0[f] + nc[f] - pc0 : 0;  // chunks from pc0 to the filter's last one
            // a row may start on any 16-byte piece; its start is pushed forward (<= 15 pieces) until its piece index mod 16
            // differs from that of every earlier row of the same slot whose lane shares one of ds_read_b128's 16-lane
            // groups -- the lanes of a group then read 16 different bank quads at every step (modelled 11.7 -> 4 cycles)
            const int grp16 = lane_group(gi);
            size_t piece = wpieces;
            for (int tries = 0; tries < 16 && (used[k][grp16] >> (piece & 15) & 1u); ++tries) ++piece;
            used[k][grp16] |= 1u << (piece & 15);
            const size_t per_chunk = 4 * wsz / 16;  // 16-byte pieces per chunk
            if (piece + size_t(own) * per_chunk > 0xFFFF) return AUD_OK;
            slots[(size_t(gi) * n_slots + k) * 2] = uint32_t(pc0) | (uint32_t(f) << 16);
            slots[(size_t(gi) * n_slots + k) * 2 + 1] = uint32_t(piece) | (uint32_t(own) << 16);
            wpieces = piece + size_t(own) * per_chunk;
            wrows.resize(wpieces * 16 / wsz, 0.0);
            wr = wrows.data() + piece * 16 / wsz;
        }
        if (hi >= lo)
            for (int bin = lo; bin <= hi; ++bin)  // x 1/4 (exact): the kernels keep FOUR times the power in LDS
                wr[bin - 4 * pc0] = 0.25 * mel_filters[int64_t(f) * (nf + 2) + (bin - lo)];
    }
    // twiddles, from the same long-double formula as the plan's W_N table
    const long double w = -2.0L * 3.14159265358979323846264338327950288L / (long double)N;
    auto tw = [&](int k, double* out) { out[0] = double(cosl(w * (k % N))); out[1] = double(sinl(w * (k % N))); };
    std::vector<double> twa, tws, gtab;
    std::vector<uint16_t> pairs;
    if (kind != 4) {
        twa.resize(size_t(g.k1_rows - 1) * g.lanes_per_frame * 2);
        tws.resize(size_t(g.split_count) * 2);
        for (int k1 = 1; k1 < g.k1_rows; ++k1)
            for (int j = 0; j < g.lanes_per_frame; ++j) tw(2 * j * k1, &twa[(size_t(k1 - 1) * g.lanes_per_frame + j) * 2]);
        for (int k = 0; k < g.split_count; ++k) tw(k, &tws[size_t(k) * 2]);
    } else {
        // w64x16 (melspec_wave.hip): pass-2 twiddles W_64^(n3 k2) = W_2048^(32 n3 k2) in the blob; the column pairs of
        // every lane; and in GLOBAL memory, lane-ordered: pass-1 twiddles W_1024^(l k1) [15][64] and the split twiddles
        // W_2048^k of the lane's 2 x 5 pairs [2][5][64]
        twa.resize(4 * 16 * 2);
        for (int n3 = 0; n3 < 4; ++n3)
            for (int k2 = 0; k2 < 16; ++k2) tw(32 * n3 * k2, &twa[(size_t(n3) * 16 + k2) * 2]);
        pairs.resize(64 * 4);
        // lane-ordered base twiddles: pass 1 needs W_1024^(l k1), k1 = 1..15 -- the kernel multiplies them together from
        // the four powers k1 = 1, 2, 4, 8 (at most three factors) -- and the split W_2048^ka of the lane's two column
        // pairs (the other pairs' twiddles are that value times an eighth root of unity)
        gtab.resize((4 * 64 + 2 * 64) * 2);
        for (int b = 0; b < 4; ++b)
            for (int l = 0; l < 64; ++l) tw(2 * l * (1 << b), &gtab[(size_t(b) * 64 + l) * 2]);
        for (int l = 0; l < 64; ++l)
            for (int sl = 0; sl < 2; ++sl) {
                // slot q < 127: columns with base bins ka = q + 1 and kb = 256 - ka (its partner column); q = 127: the two
                // self-paired columns 128 and 0
                const int q = l + 64 * sl;
                const bool sp = q == 127;
                const int ka = sp ? 128 : q + 1, kb = sp ? 0 : 255 - q;
                pairs[4 * l + 2 * sl] = uint16_t(ka);
                pairs[4 * l + 2 * sl + 1] = uint16_t(kb);
                tw(ka, &gtab[(size_t(4 * 64) + size_t(sl) * 64 + l) * 2]);
            }
    }
    // the blob
    auto align32 = [](size_t v) { return (v + 31) & ~size_t(31); };
    const size_t w4_bytes = align32(wrows.size() * wsz);
    e.w4_off = 0;
    e.w_stride = int(w_stride);
    e.slots_off = int(w4_bytes);
    e.n_slots = n_slots;
    e.twa_off = int(e.slots_off + align32(slots.size() * 4));
    e.tws_off = int(e.twa_off + align32(twa.size() * tsz));
    e.pairs_off = int(e.tws_off + align32(tws.size() * tsz));
    e.blob_bytes = int(e.pairs_off + align32(pairs.size() * 2));
    e.n_groups = G;
    // fused segment tail (w16x16 / w20x10, up to kDctCoefs coefficients): the transposed DCT-I rows, one row per filter
    // (the same formula as the plan's d_dct table, capi.hip)
    std::vector<double> dct_t;
    e.dct_off = -1;
    if (kind != 4 && d.mfcc_coefs >= 1 && d.mfcc_coefs <= aud::kDctCoefs && d.segment_steps <= N / 2 + 1) {
        dct_t.assign(size_t(nf) * aud::kDctPitch, 0.0);
        const long double pi = 3.14159265358979323846264338327950288L;
        for (int k = 0; k < d.mfcc_coefs; ++k)
            for (int j = 0; j < nf; ++j)
                dct_t[size_t(j) * aud::kDctPitch + k] =
                    j == 0 ? 1.0 : j == nf - 1 ? ((k & 1) ? -1.0 : 1.0)
                                               : double(2.0L * cosl(pi * (long double)j * (long double)k / (long double)(nf - 1)));
        e.dct_off = e.blob_bytes;
        e.blob_bytes += int(align32(dct_t.size() * tsz));
    }
    std::vector<unsigned char> blob(size_t(e.blob_bytes), 0);
    auto put_real = [&](size_t off, const std::vector<double>& v) {
        if (v.empty()) return;
        if (dt == AUD_F64) std::memcpy(&blob[off], v.data(), v.size() * 8);
        else {
            std::vector<float> fv = convert<float>(v.data(), v.size());
            std::memcpy(&blob[off], fv.data(), fv.size() * 4);
        }
    };
    {
        std::vector<float> fw = convert<float>(wrows.data(), wrows.size());
        std::memcpy(&blob[size_t(e.w4_off)], fw.data(), fw.size() * 4);
    }
    std::memcpy(&blob[size_t(e.slots_off)], slots.data(), slots.size() * 4);
    put_real(size_t(e.twa_off), twa);
    put_real(size_t(e.tws_off), tws);
    if (!pairs.empty()) std::memcpy(&blob[size_t(e.pairs_off)], pairs.data(), pairs.size() * 2);
    if (e.dct_off >= 0) put_real(size_t(e.dct_off), dct_t);
    if (!aud::melspec_wave_finish(kind, dt, &e)) return AUD_OK;  // does not fit LDS
    if (aud::melspec_wave_prepare(kind, dt, &e) != hipSuccess) {
        (void)hipGetLastError();
        return AUD_OK;
    }
    int rc = upload(c, &p->d_blob, blob.data(), blob.size());
    if (rc != AUD_OK) return rc;
    if (!gtab.empty()) {
        if (dt == AUD_F64) rc = upload(c, &p->d_gtab, gtab.data(), gtab.size() * 8);
        else {
            std::vector<float> fv = convert<float>(gtab.data(), gtab.size());
            rc = upload(c, &p->d_gtab, fv.data(), fv.size() * 4);
        }
        if (rc != AUD_OK) return rc;
    }
    e.gtab = p->d_gtab;
    e.blob = p->d_blob;
    p->wv = e;
    p->wave_kind = kind;
    p->use_wave = true;
    p->family = plan_family(p);
    return AUD_OK;
}

}  // namespace audc
