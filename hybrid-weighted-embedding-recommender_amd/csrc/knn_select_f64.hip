// Stage 2 of the exact k-NN on GENUINE float64 rows (round 6): the float64-row instantiations of the select / re-rank / exhaustive-pass templates
// of knn_select_kernels.h and their launchers.  The reference keeps ContentRecommendation's `vectors` in float64 (hwer/content_recommender.py:96,
// hwer/recommendation_base.py:109); its KD-tree (:74, :79) and predict (:141-143) work on those rows.  The MFMA screen runs on the float32 / float16
// rounding of the table (a screen only has to keep a superset; the select stage widens its error bound by the rounding, e32_bound<true>), every VALUE -
// the float64 distances in scikit-learn's order, the scores in numpy's pairwise order - is computed from the float64 rows inside the SAME fused
// kernels the float32 flavour runs: no candidate lists through HBM, no separate distance / ordering / scoring launches (rounds 3-5: ~12 launches).
// A translation unit of its own so that the two flavours compile side by side.
#include "knn_select_kernels.h"

namespace hwer {

int launch_select_list_f64(const DenseSelectArgs& da, int d, int score_mode, bool coop, hipStream_t st) {
    const size_t lds = coop ? coop_lds_bytes(d, da.kp) : size_t(4) * wave_lds_bytes(d, da.kp);
    auto launch = [&](auto md) -> int {
        constexpr int MD = decltype(md)::value;
        if (coop) {
            if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<16, MD, true, true, true>), lds, "knn_select_list_f64")) return e;
            hipLaunchKernelGGL((select_dense_kernel<16, MD, true, true, true>), dim3(unsigned(da.B)), dim3(256), lds, st, da);
            return HWER_OK;
        }
        if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<16, MD, true, false, true>), lds, "knn_select_list_f64")) return e;
        hipLaunchKernelGGL((select_dense_kernel<16, MD, true, false, true>), dim3(unsigned(cdiv64(da.B, 4))), dim3(256), lds, st, da);
        return HWER_OK;
    };
    const int rc = score_mode == 0 ? launch(std::integral_constant<int, 0>{}) : launch(std::integral_constant<int, 2>{});
    if (rc) return rc;
    return check_launch("knn_select_list_f64");
}

int launch_select_dense_f64(const DenseSelectArgs& da, int d, int score_mode, bool coop, hipStream_t st) {
    const size_t lds = coop ? coop_lds_bytes(d, da.kp) : size_t(4) * wave_lds_bytes(d, da.kp);
    auto launch2 = [&](auto mv, auto md) -> int {
        constexpr int MV = decltype(mv)::value, MD = decltype(md)::value;
        if (coop) {
            if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<MV, MD, false, true, true>), lds, "knn_select_dense_f64")) return e;
            hipLaunchKernelGGL((select_dense_kernel<MV, MD, false, true, true>), dim3(unsigned(da.B)), dim3(256), lds, st, da);
            return HWER_OK;
        }
        if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<MV, MD, false, false, true>), lds, "knn_select_dense_f64")) return e;
        hipLaunchKernelGGL((select_dense_kernel<MV, MD, false, false, true>), dim3(unsigned(cdiv64(da.B, 4))), dim3(256), lds, st, da);
        return HWER_OK;
    };
    auto launch = [&](auto mv) -> int {
        return score_mode == 0 ? launch2(mv, std::integral_constant<int, 0>{}) : launch2(mv, std::integral_constant<int, 2>{});
    };
    int rc;
    if (da.ld <= 256 * 4) rc = launch(std::integral_constant<int, 4>{});
    else if (da.ld <= 256 * 8) rc = launch(std::integral_constant<int, 8>{});
    else if (da.ld <= 256 * 16) rc = launch(std::integral_constant<int, 16>{});
    else rc = launch(std::integral_constant<int, 32>{});
    if (rc) return rc;
    return check_launch("knn_select_dense_f64");
}

int launch_select_rerank_f64(const SelectArgs& sa, int d, int score_mode, hipStream_t st) {
    const size_t lds = size_t(4) * wave_lds_bytes(d, sa.kcap);
    auto launch = [&](auto md) -> int {
        constexpr int MD = decltype(md)::value;
        if (int e = set_max_lds(reinterpret_cast<const void*>(&select_rerank_kernel<MD, true>), lds, "knn_select_f64")) return e;
        hipLaunchKernelGGL((select_rerank_kernel<MD, true>), dim3(unsigned(cdiv64(sa.B, 4))), dim3(256), lds, st, sa);
        return HWER_OK;
    };
    const int rc = score_mode == 0 ? launch(std::integral_constant<int, 0>{}) : launch(std::integral_constant<int, 2>{});
    if (rc) return rc;
    return check_launch("knn_select_f64");
}

int launch_exact_device_f64(const ExactDevArgs& ea, size_t lds, hipStream_t st) {
    if (int e = set_max_lds(reinterpret_cast<const void*>(&exact_device_kernel<true>), lds, "knn_exact_device_f64")) return e;
    hipLaunchKernelGGL(exact_device_kernel<true>, dim3(unsigned(ea.n_chunks), EXACT_DEVICE_MAXQ), dim3(256), lds, st, ea);
    return check_launch("knn_exact_device_f64");
}

}  // namespace hwer
