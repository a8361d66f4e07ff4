// torch.ops.lerf.* -- dispatcher registration (TORCH_LIBRARY) of the hot path, SURVEY.md 8(b) "torch custom-op layer".
//
// Host C++ only: no device code here.  Every op validates its tensors (TORCH_CHECK -> Python RuntimeError), allocates
// its outputs with at::empty, and enqueues the kernels of liblerf_hip.so through the C ABI (include/lerf_hip.h) on
// PyTorch's current HIP stream; nothing synchronises with the host.  The implementations are registered under the
// CUDA dispatch key, which is the key of the HIP backend in PyTorch-ROCm; there is deliberately NO CPU kernel: a CPU
// tensor fails in the dispatcher ("no kernel for CPU"), it does not fall back.
//
// What the ops stand in for in the reference:
//   lerf::lut_stages    stages 1+2 of eltr._worker                 resample/eval_lut_sr.py:541-628
//   lerf::sr_fused      the whole SR path, uint8 in -> uint8 out   resample/eval_lut_sr.py:541-665
//   lerf::warp_fused    the homographic warp path                  resample/eval_lut_warp.py:100-222
//   lerf::resize_gauss  SteeringGaussianResize2dTorch.resize       resize_right/resize_right2d_torch.py:154-197
//   lerf::resize_linear AmplifiedLinearResize2dTorch.resize        resize_right/resize_right2d_torch.py:214-247
//   lerf::resize_backward  what autograd derives for the two above (train_model.py:431-441); the autograd formulas
//                       themselves are attached in lerf_pytorch_amd/torch_ops.py (torch.library.register_autograd)
// LUT sets travel as tensor lists: luts_s1 = [s, c, t] int8 [17^4, 1]; luts_s2 = [s_r0, s_r1, c_r0, c_r1, t_r0, t_r1]
// int8 [17^4, outC]; pack = the fused LUT pack (lerf_fused_lutpack_build) or None.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "lerf_hip.h"

namespace {

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

// every op runs on the device of its first tensor: the current stream and at::empty's allocations are that device's
using DeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == LERF_OK, what, ": ", lerf_strerror(rc)); }

struct LutView {
    lerf_luts_t st{};
    std::vector<at::Tensor> keep;
};

LutView make_luts(at::TensorList s1, at::TensorList s2, const c10::optional<at::Tensor>& pack, const at::Device& dev) {
    TORCH_CHECK(s1.size() == 3 && s2.size() == 6, "luts_s1 = [s, c, t], luts_s2 = [s_r0, s_r1, c_r0, c_r1, t_r0, t_r1]");
    LutView v;
    v.st.n_modes1 = 3;
    v.st.n_modes2 = 3;
    memcpy(v.st.modes1, "sct", 3);
    memcpy(v.st.modes2, "sct", 3);
    const int64_t oC = s2[0].numel() / LERF_LUT_ENTRIES;
    TORCH_CHECK(oC == 1 || oC == 3, "stage-2 LUTs must be [17^4, 1] (LeRF-L) or [17^4, 3] (LeRF-G)");
    v.st.oC = (int)oC;
    for (int m = 0; m < 3; ++m) {
        TORCH_CHECK(s1[m].scalar_type() == at::kChar && s1[m].numel() == LERF_LUT_ENTRIES && s1[m].device() == dev,
                    "stage-1 LUTs must be int8 [17^4] on the image's device");
        v.keep.push_back(s1[m].contiguous());
        v.st.s1[m] = (const int8_t*)v.keep.back().data_ptr();
        for (int r = 0; r < 2; ++r) {
            const at::Tensor& t = s2[2 * m + r];
            TORCH_CHECK(t.scalar_type() == at::kChar && t.numel() == LERF_LUT_ENTRIES * oC && t.device() == dev,
                        "stage-2 LUTs must be int8 [17^4, outC] on the image's device");
            v.keep.push_back(t.contiguous());
            v.st.s2[m][r] = (const int8_t*)v.keep.back().data_ptr();
        }
    }
    v.st.fused_pack = nullptr;
    if (pack.has_value() && pack->defined()) {
        TORCH_CHECK(pack->scalar_type() == at::kByte && (size_t)pack->numel() >= lerf_fused_lutpack_bytes(&v.st) && pack->device() == dev &&
                        pack->is_contiguous(),
                    "pack must be the uint8 buffer lerf_fused_lutpack_build filled");
        v.st.fused_pack = pack->data_ptr();
    }
    return v;
}

// separable SR geometry, cached per (size, scale, support, arithmetic, device): two small tables per axis
struct SrGeo {
    at::Tensor left_r, dis_r, left_c, dis_c, dis_r64, dis_c64;
    lerf_sr_geo_t st{};
    int oH = 0, oW = 0;
};

const SrGeo& sr_geometry(int H, int W, double sh, double sw, int S, bool torch32, const at::Device& dev) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, double, double, int, bool, int>, SrGeo> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(H, W, sh, sw, S, torch32, (int)dev.index());
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    TORCH_CHECK(sh > 0 && sw > 0 && S >= 1 && S <= LERF_MAX_SUPPORT, "bad scale / support");
    SrGeo g;
    g.oH = lerf_out_size(H, sh);
    g.oW = lerf_out_size(W, sw);
    auto axis = [&](int n_in, int n_out, double s, at::Tensor& left, at::Tensor& d32, at::Tensor& d64) {
        at::Tensor l = at::empty({n_out}, at::kInt), f = at::empty({n_out, S}, at::kFloat), d = at::empty({n_out, S}, at::kDouble);
        int32_t pads[2];
        if (torch32) {
            check_rc(lerf_sr_axis_tables_f32(n_in, n_out, s, S, l.data_ptr<int32_t>(), f.data_ptr<float>(), pads), "lerf_sr_axis_tables_f32");
            d = f.to(at::kDouble);
        } else {
            check_rc(lerf_sr_axis_tables(n_in, n_out, s, S, l.data_ptr<int32_t>(), d.data_ptr<double>(), f.data_ptr<float>(), pads),
                     "lerf_sr_axis_tables");
        }
        left = l.to(dev);
        d32 = f.to(dev);
        d64 = d.to(dev);
    };
    axis(H, g.oH, sh, g.left_r, g.dis_r, g.dis_r64);
    axis(W, g.oW, sw, g.left_c, g.dis_c, g.dis_c64);
    g.st.S = S;
    g.st.out_h = g.oH;
    g.st.out_w = g.oW;
    g.st.left_r = g.left_r.data_ptr<int32_t>();
    g.st.dis_r = g.dis_r.data_ptr<float>();
    g.st.left_c = g.left_c.data_ptr<int32_t>();
    g.st.dis_c = g.dis_c.data_ptr<float>();
    g.st.dis_r64 = g.dis_r64.data_ptr<double>();
    g.st.dis_c64 = g.dis_c64.data_ptr<double>();
    g.st.pad_mode = LERF_PAD_CONSTANT;
    return cache.emplace(key, std::move(g)).first->second;
}

void check_u8_frames(const at::Tensor& img) {
    TORCH_CHECK(img.scalar_type() == at::kByte && (img.dim() == 3 || img.dim() == 4), "img must be uint8 [H,W,C] or [N,H,W,C]");
}

// ------------------------------------------------------------------------------------------------ ops
std::tuple<at::Tensor, at::Tensor> lut_stages(const at::Tensor& img, at::TensorList s1, at::TensorList s2) {
    TORCH_CHECK(img.scalar_type() == at::kByte && img.dim() == 3, "img must be uint8 [H,W,C]");
    DeviceGuard guard(img.device());
    at::Tensor x = img.contiguous();
    LutView L = make_luts(s1, s2, c10::nullopt, x.device());
    const int H = (int)x.size(0), W = (int)x.size(1), C = (int)x.size(2);
    at::Tensor feat = at::empty_like(x), hq = at::empty({H, W, C, L.st.oC}, x.options());
    lerf_plane_t pi{x.data_ptr(), LERF_U8, (int64_t)W * C, C, 1};
    lerf_mplane_t pf{feat.data_ptr(), LERF_U8, (int64_t)W * C, C, 1};
    lerf_mplane_t ph{hq.data_ptr(), LERF_U8, (int64_t)W * C * L.st.oC, (int64_t)C * L.st.oC, L.st.oC};
    check_rc(lerf_lut_stages_u8(&pi, H, W, C, &L.st, &pf, &ph, cur_stream()), "lerf_lut_stages_u8");
    return {feat, hq};
}

at::Tensor sr_fused(const at::Tensor& img, at::TensorList s1, at::TensorList s2, const c10::optional<at::Tensor>& pack, double scale_h,
                    double scale_w, int64_t support, double max_sigma) {
    check_u8_frames(img);
    DeviceGuard guard(img.device());
    const bool squeeze = img.dim() == 3;
    at::Tensor x = (squeeze ? img.unsqueeze(0) : img).contiguous();
    LutView L = make_luts(s1, s2, pack, x.device());
    const bool linear = L.st.oC == 1;                                    // the harness builds the linear resizer with its defaults
    const int S = linear ? 2 : (int)support;                             // (eval_lut_sr.py:482-484): S = 2, max_sigma = 1
    const double ms = linear ? 1.0 : max_sigma;
    const int N = (int)x.size(0), H = (int)x.size(1), W = (int)x.size(2), C = (int)x.size(3);
    const SrGeo& g = sr_geometry(H, W, scale_h, scale_w, S, false, x.device());
    at::Tensor out = at::empty({N, g.oH, g.oW, C}, x.options());
    at::Tensor ws = at::empty({(int64_t)std::max<size_t>(lerf_sr_fused_workspace_bytes(H, W, C, N), 1)}, x.options());
    check_rc(lerf_sr_fused_u8((const uint8_t*)x.data_ptr(), x.stride(0), N, H, W, C, &L.st, &g.st, linear ? LERF_KIND_LINEAR : LERF_KIND_GAUSS,
                              ms, (uint8_t*)out.data_ptr(), out.stride(0), ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
             "lerf_sr_fused_u8");
    return squeeze ? out.squeeze(0) : out;
}

at::Tensor warp_fused(const at::Tensor& img, at::TensorList s1, at::TensorList s2, const c10::optional<at::Tensor>& pack,
                      const at::Tensor& matrix, int64_t out_h, int64_t out_w, int64_t support, double max_sigma) {
    TORCH_CHECK(img.scalar_type() == at::kByte && img.dim() == 3, "img must be uint8 [H,W,C]");
    TORCH_CHECK(matrix.numel() == 9, "matrix must be 3x3 (input -> output coordinates)");
    DeviceGuard guard(img.device());
    at::Tensor x = img.contiguous();
    LutView L = make_luts(s1, s2, pack, x.device());
    const bool linear = L.st.oC == 1;
    const int H = (int)x.size(0), W = (int)x.size(1), C = (int)x.size(2);
    lerf_warp_geo_t g{};
    g.S = linear ? 2 : (int)support;
    g.out_h = (int)out_h;
    g.out_w = (int)out_w;
    at::Tensor minv = at::linalg_inv(matrix.detach().to(at::kCPU, at::kDouble).reshape({3, 3})).contiguous();   // np.linalg.inv (:327)
    memcpy(g.minv, minv.data_ptr<double>(), sizeof(g.minv));
    int32_t pads[4];
    check_rc(lerf_warp_pads(g.minv, H, W, g.out_h, g.out_w, g.S, pads), "lerf_warp_pads");
    g.pad_r_lo = pads[0]; g.pad_r_hi = pads[1]; g.pad_c_lo = pads[2]; g.pad_c_hi = pads[3];
    g.pad_mode = LERF_PAD_CONSTANT;
    const int kind = linear ? LERF_KIND_LINEAR : LERF_KIND_GAUSS;
    const double ms = linear ? 1.0 : max_sigma;
    at::Tensor out = at::empty({out_h, out_w, C}, x.options());
    lerf_mplane_t po{out.data_ptr(), LERF_U8, (int64_t)out_w * C, C, 1};
    if (C == 3 && L.st.fused_pack) {
        at::Tensor packed = at::empty({H, W, C}, x.options().dtype(at::kInt));
        at::Tensor ws = at::empty({(int64_t)std::max<size_t>(lerf_sr_fused_workspace_bytes(H, W, C, 1), 1)}, x.options());
        check_rc(lerf_stages_packed_u8((const uint8_t*)x.data_ptr(), 0, 1, H, W, C, &L.st, (uint32_t*)packed.data_ptr(), 0, ws.data_ptr(),
                                       (size_t)ws.numel(), cur_stream()),
                 "lerf_stages_packed_u8");
        check_rc(lerf_warp_packed((const uint32_t*)packed.data_ptr(), 0, 1, H, W, C, &g, kind, ms, &po, 0, cur_stream()), "lerf_warp_packed");
        return out;
    }
    auto fh = lut_stages(x, s1, s2);
    const at::Tensor &feat = std::get<0>(fh), &hq = std::get<1>(fh);
    lerf_plane_t pf{feat.data_ptr(), LERF_U8, (int64_t)W * C, C, 1};
    lerf_plane_t hp[3];
    for (int k = 0; k < 3; ++k)
        hp[k] = lerf_plane_t{(const uint8_t*)hq.data_ptr() + (k < L.st.oC ? k : 0), LERF_U8, (int64_t)W * C * L.st.oC, (int64_t)C * L.st.oC,
                             L.st.oC};
    check_rc(lerf_warp(&pf, hp, H, W, C, &g, kind, ms, &po, cur_stream()), "lerf_warp");
    return out;
}

// planar float32 maps [B,C,H,W] -> [B,C,oH,oW]; float32 geometry of the torch classes (resize_right2d_torch.py:48-103)
at::Tensor resize_planar(int kind, const at::Tensor& feat, const std::vector<at::Tensor>& hs, double sh, double sw, int64_t S, double ms) {
    TORCH_CHECK(feat.dim() == 4 && feat.scalar_type() == at::kFloat, "feat must be float32 [B,C,H,W]");
    DeviceGuard guard(feat.device());
    const int B = (int)feat.size(0), C = (int)feat.size(1), H = (int)feat.size(2), W = (int)feat.size(3);
    at::Tensor x = feat.contiguous();
    std::vector<at::Tensor> h;
    for (const at::Tensor& t : hs) {
        TORCH_CHECK(t.sizes() == feat.sizes() && t.scalar_type() == at::kFloat && t.device() == feat.device(),
                    "hyper-parameter maps must match feat (float32, same shape and device)");
        h.push_back(t.contiguous());
    }
    const SrGeo& g = sr_geometry(H, W, sh, sw, (int)S, true, x.device());
    at::Tensor out = at::empty({B, C, g.oH, g.oW}, x.options());
    const int64_t hw = (int64_t)H * W;
    lerf_plane_t pf{x.data_ptr(), LERF_F32, W, 1, hw};
    lerf_plane_t hp[3];
    for (int k = 0; k < 3; ++k) hp[k] = lerf_plane_t{h[k < (int)h.size() ? k : 0].data_ptr(), LERF_F32, W, 1, hw};
    lerf_mplane_t po{out.data_ptr(), LERF_F32, g.oW, 1, (int64_t)g.oH * g.oW};
    check_rc(lerf_resize(&pf, hp, H, W, B * C, &g.st, kind, ms, &po, cur_stream()), "lerf_resize");
    return out;
}

at::Tensor resize_gauss(const at::Tensor& feat, const at::Tensor& rho, const at::Tensor& sigma_x, const at::Tensor& sigma_y, double scale_h,
                        double scale_w, int64_t support, double max_sigma) {
    return resize_planar(LERF_KIND_GAUSS, feat, {rho, sigma_x, sigma_y}, scale_h, scale_w, support, max_sigma);
}

at::Tensor resize_linear(const at::Tensor& feat, const at::Tensor& alpha, double scale_h, double scale_w, double max_sigma) {
    return resize_planar(LERF_KIND_LINEAR, feat, {alpha}, scale_h, scale_w, 2, max_sigma);
}

// gradients of resize_gauss (kind 0: feat, rho, sigma_x, sigma_y) / resize_linear (kind 1: feat, alpha; the last two are zeros)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> resize_backward(int64_t kind, const at::Tensor& grad_out, const at::Tensor& feat,
                                                                           const at::Tensor& h0, const at::Tensor& h1, const at::Tensor& h2,
                                                                           double scale_h, double scale_w, int64_t support, double max_sigma) {
    TORCH_CHECK(kind == LERF_KIND_GAUSS || kind == LERF_KIND_LINEAR, "kind: 0 = gauss, 1 = linear");
    TORCH_CHECK(feat.dim() == 4 && feat.scalar_type() == at::kFloat && grad_out.scalar_type() == at::kFloat, "float32 [B,C,H,W] tensors");
    DeviceGuard guard(feat.device());
    const int B = (int)feat.size(0), C = (int)feat.size(1), H = (int)feat.size(2), W = (int)feat.size(3);
    const SrGeo& g = sr_geometry(H, W, scale_h, scale_w, (int)support, true, feat.device());
    TORCH_CHECK(grad_out.dim() == 4 && grad_out.size(0) == B && grad_out.size(1) == C && grad_out.size(2) == g.oH && grad_out.size(3) == g.oW &&
                    grad_out.device() == feat.device(),
                "grad_out must be [B,C,oH,oW] of the output geometry, on feat's device");
    for (const at::Tensor* t : {&h0, &h1, &h2})         // all three are read by the Gaussian kernel, h0 by the linear one
        TORCH_CHECK((t != &h0 && kind == LERF_KIND_LINEAR) ||
                        (t->sizes() == feat.sizes() && t->scalar_type() == at::kFloat && t->device() == feat.device()),
                    "hyper-parameter maps must match feat (float32, same shape and device)");
    at::Tensor x = feat.contiguous(), go = grad_out.contiguous(), a = h0.contiguous(), b = h1.contiguous(), c = h2.contiguous();
    at::Tensor gx = at::zeros_like(x), g0 = at::zeros_like(x), g1 = at::zeros_like(x), g2 = at::zeros_like(x);
    const bool gauss = kind == LERF_KIND_GAUSS;
    check_rc(lerf_resize_bwd_f32(x.data_ptr<float>(), a.data_ptr<float>(), gauss ? b.data_ptr<float>() : nullptr,
                                 gauss ? c.data_ptr<float>() : nullptr, B * C, H, W, &g.st, (int)kind, max_sigma, go.data_ptr<float>(),
                                 gx.data_ptr<float>(), g0.data_ptr<float>(), gauss ? g1.data_ptr<float>() : nullptr,
                                 gauss ? g2.data_ptr<float>() : nullptr, cur_stream()),
             "lerf_resize_bwd_f32");
    return {gx, g0, g1, g2};
}

}  // namespace

TORCH_LIBRARY(lerf, m) {
    m.def("lut_stages(Tensor img, Tensor[] luts_s1, Tensor[] luts_s2) -> (Tensor, Tensor)");
    m.def("sr_fused(Tensor img, Tensor[] luts_s1, Tensor[] luts_s2, Tensor? pack, float scale_h, float scale_w, int support, float max_sigma) -> Tensor");
    m.def("warp_fused(Tensor img, Tensor[] luts_s1, Tensor[] luts_s2, Tensor? pack, Tensor matrix, int out_h, int out_w, int support, float max_sigma) -> Tensor");
    m.def("resize_gauss(Tensor feat, Tensor rho, Tensor sigma_x, Tensor sigma_y, float scale_h, float scale_w, int support, float max_sigma) -> Tensor");
    m.def("resize_linear(Tensor feat, Tensor alpha, float scale_h, float scale_w, float max_sigma) -> Tensor");
    m.def("resize_backward(int kind, Tensor grad_out, Tensor feat, Tensor h0, Tensor h1, Tensor h2, float scale_h, float scale_w, int support, float max_sigma) -> (Tensor, Tensor, Tensor, Tensor)");
}

// PyTorch-ROCm dispatches HIP tensors under the key named CUDA
TORCH_LIBRARY_IMPL(lerf, CUDA, m) {
    m.impl("lut_stages", lut_stages);
    m.impl("sr_fused", sr_fused);
    m.impl("warp_fused", warp_fused);
    m.impl("resize_gauss", resize_gauss);
    m.impl("resize_linear", resize_linear);
    m.impl("resize_backward", resize_backward);
}
