// Host-only check of the ragged-frame plan of the Gram launch (ccal::gram2_bin_plan, csrc/ccal_kernels_gram2.hip; declarations
// repeated from csrc/ccal_internal.hpp - plain data, no HIP): every frame exactly once, bins contiguous in the order sorted by
// corner count (or ONE bin over the FOLDED order), every frame covered by its bin's lanes within the launch's trip-count limit, workgroup ranges consistent;
// uniform frames and small problems are left alone.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ccal {
constexpr int kGramMaxBins = 5;
struct GramBins {
    int32_t n_bins = 0;
    int32_t lpf[kGramMaxBins] = {}, first[kGramMaxBins] = {}, count[kGramMaxBins] = {}, wg0[kGramMaxBins + 1] = {};
    int32_t fold = 0;
};
GramBins gram2_bin_plan(const int64_t* obs_off, int n_obs, bool two_per_simd, std::vector<int32_t>* order, bool rig_list = false);
}  // namespace ccal

static int fail(const char* what) { std::printf("PLAN-FAIL %s\n", what); return 1; }

static int check(const std::vector<int64_t>& off, bool two, bool expect_bins, const char* label) {
    const int n_obs = (int)off.size() - 1;
    std::vector<int32_t> order;
    const ccal::GramBins gb = ccal::gram2_bin_plan(off.data(), n_obs, two, &order);
    if (!expect_bins) return gb.n_bins == 0 ? 0 : fail(label);
    if (gb.n_bins < 1 || gb.n_bins > ccal::kGramMaxBins) return fail("bin count");
    if ((int)order.size() != n_obs) return fail("order size");
    std::vector<char> seen((size_t)n_obs, 0);
    for (int32_t o : order) { if (o < 0 || o >= n_obs || seen[(size_t)o]) return fail("order is not a permutation"); seen[(size_t)o] = 1; }
    auto cnt = [&](int pos) { return off[order[(size_t)pos] + 1] - off[order[(size_t)pos]]; };
    if (gb.fold > 0) {
        // ONE bin, the table folded: the larger half largest first, the smaller half smallest first; the fold sits where the single-launch
        // groups split their workgroups (four wavefronts of eight)
        const int g = 64 / gb.lpf[0], waves = (n_obs + g - 1) / g;
        if (gb.n_bins != 1 || !two) return fail("a folded plan has one bin (two wavefronts per SIMD)");
        if (gb.fold != std::min(n_obs, 4 * ((waves + 7) / 8) * g)) return fail("fold position");
        for (int i = 1; i < gb.fold; ++i) if (cnt(i) > cnt(i - 1)) return fail("front half not sorted largest first");
        for (int i = gb.fold + 1; i < n_obs; ++i) if (cnt(i) < cnt(i - 1)) return fail("back half not sorted smallest first");
        if (gb.fold < n_obs && cnt(n_obs - 1) > cnt(gb.fold - 1)) return fail("the halves overlap");
        if (gb.first[0] != 0 || gb.count[0] != n_obs || gb.wg0[0] != 0 || gb.wg0[1] != (waves + 1) / 2) return fail("folded bin range");
        std::printf("%s: folded at %d, %d lanes x %d frames, %d workgroups, trip counts %d .. %d\n", label, gb.fold, gb.lpf[0], n_obs, gb.wg0[1],
                    (int)((cnt(gb.fold) + gb.lpf[0] - 1) / gb.lpf[0]), (int)((cnt(0) + gb.lpf[0] - 1) / gb.lpf[0]));
        return 0;
    }
    for (int i = 1; i < n_obs; ++i) if (cnt(i) > cnt(i - 1)) return fail("not sorted by corner count");
    int pos = 0, wgs = 0, T = 0;
    for (int b = 0; b < gb.n_bins; ++b) {
        if (gb.first[b] != pos || gb.count[b] <= 0 || gb.wg0[b] != wgs) return fail("bin ranges");
        const int lpf = gb.lpf[b], g = 64 / lpf;
        if (lpf != 6 && lpf != 8 && lpf != 12 && lpf != 16 && lpf != 32) return fail("lanes per frame");
        if (b > 0 && lpf >= gb.lpf[b - 1]) return fail("bins in launch order: wider first");
        T = std::max<int>(T, (int)((cnt(pos) + lpf - 1) / lpf));
        wgs += ((gb.count[b] + g - 1) / g + 1) / 2;
        pos += gb.count[b];
    }
    if (pos != n_obs || gb.wg0[gb.n_bins] != wgs) return fail("bins do not cover the frames");
    // a frame sits in the NARROWEST bin that covers it within the launch's trip-count limit
    for (int b = 0; b + 1 < gb.n_bins; ++b)
        for (int q = gb.first[b]; q < gb.first[b] + gb.count[b]; ++q)
            if (cnt(q) <= (int64_t)T * gb.lpf[b + 1] && q > gb.first[b]) { /* boundary frames may stay with their group */ if (cnt(q) <= (int64_t)(T - 1) * gb.lpf[b + 1]) return fail("a frame sits in too wide a bin"); }
    std::printf("%s: %d bins, T = %d, %d workgroups:", label, gb.n_bins, T, wgs);
    for (int b = 0; b < gb.n_bins; ++b) std::printf(" %d lanes x %d frames", gb.lpf[b], gb.count[b]);
    std::printf("\n");
    return 0;
}

int main() {
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int lo, int hi) { s = s * 6364136223846793005ull + 1442695040888963407ull; return lo + (int)((s >> 33) % (uint64_t)(hi - lo + 1)); };
    auto make = [&](int n_obs, int lo, int hi) { std::vector<int64_t> off((size_t)n_obs + 1, 0); for (int i = 0; i < n_obs; ++i) off[(size_t)i + 1] = off[(size_t)i] + rnd(lo, hi); return off; };
    int bad = 0;
    bad += check(make(10000, 24, 144), true, true, "10000 frames U{24..144}, two wavefronts per SIMD");
    bad += check(make(50000, 24, 144), true, true, "50000 frames U{24..144}");
    bad += check(make(50000, 24, 144), false, true, "50000 frames U{24..144}, one wavefront per SIMD");
    bad += check(make(3000, 6, 700), true, true, "3000 frames U{6..700}");
    bad += check(make(10000, 144, 144), true, false, "uniform frames are not binned");
    bad += check(make(1500, 24, 144), true, false, "fewer than 2000 frames are not binned");
    {   // two sizes only
        std::vector<int64_t> off(1, 0);
        for (int i = 0; i < 6000; ++i) off.push_back(off.back() + (i % 3 == 0 ? 144 : 36));
        bad += check(off, true, true, "6000 frames of 144 or 36 corners");
    }
    if (!bad) std::printf("PLAN-OK\n");
    return bad ? 1 : 0;
}
