// pybind11 / torch shim over the C ABI of include/egoprior.h, in the convention of the reference's native modules:
// one `pybind_<name>_ext(py::module&)` per directory under csrc/ (registered from csrc/bind.cpp:20-49), functions that take
// and return torch::Tensor, CHECK_CONTIGUOUS on the inputs (csrc/droid_net_ext/droid.cpp:10-11), outputs allocated by the
// callee (csrc/slam_ext/geom_kernels.cu:1497), errors as C++ exceptions (TORCH_CHECK).  Unlike the reference's modules it
// launches on the current stream of the tensors' device and releases the GIL around the call, so that a decode thread can
// run while a frame is issued.  The arithmetic is all behind the C ABI: this file only translates types.
#include <torch/extension.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <vector>

#include "egoprior.h"

#define CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define CHECK_DEVICE(x) TORCH_CHECK((x).is_cuda(), #x " must be a device tensor")
#define CHECK_INPUT(x) \
    CHECK_DEVICE(x);   \
    CHECK_CONTIGUOUS(x)

namespace {

void check(int rc) { TORCH_CHECK(rc == 0, "egoprior: ", egoprior_last_error()); }

void* current_stream(const torch::Tensor& t) {
    // (a ROCm build of torch calls its devices "cuda": the stream and guard types that go with that name)
    return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

// rows of a (3|4, 4) host matrix, row-major 3x4
template <typename T>
void copy_3x4(const torch::Tensor& m, T* out, const char* what) {
    TORCH_CHECK(!m.is_cuda() && m.dim() == 2 && m.size(1) == 4 && (m.size(0) == 3 || m.size(0) == 4), what,
                " must be a (3,4) or (4,4) host tensor");
    const torch::Tensor c = m.to(std::is_same<T, double>::value ? torch::kFloat64 : torch::kFloat32).contiguous();
    const T* p = c.data_ptr<T>();
    for (int i = 0; i < 12; ++i) out[i] = p[i];
}

// scripts/render_vipe_pointcloud.py:358-430 / :682-751 / :594-654: one exo frame -> compacted world points.
// Returns (points (n_s, 4) float32 records: x, y, z and the colour bytes in the 4th word; count (1,) int32; flags (1,)
// int32: some colour byte > 1).  depth float32 or float16 (H, W), rgb uint8 (H, W, 3), mask uint8 (H, W) or None.
std::vector<torch::Tensor> unproject_frame(torch::Tensor depth, torch::Tensor rgb, c10::optional<torch::Tensor> mask,
                                           std::vector<double> intrinsics, torch::Tensor c2w, torch::Tensor T_world,
                                           int64_t subsample, int64_t select) {
    CHECK_INPUT(depth);
    CHECK_INPUT(rgb);
    TORCH_CHECK(depth.dim() == 2 && rgb.dim() == 3 && rgb.size(2) == 3 && rgb.size(0) == depth.size(0) &&
                    rgb.size(1) == depth.size(1) && rgb.scalar_type() == torch::kUInt8,
                "depth (H,W) and rgb (H,W,3) uint8 expected");
    TORCH_CHECK(depth.scalar_type() == torch::kFloat32 || depth.scalar_type() == torch::kFloat16, "depth: float32 or float16");
    TORCH_CHECK(intrinsics.size() == 4, "intrinsics = [fx, fy, cx, cy]");
    TORCH_CHECK(subsample >= 1, "subsample >= 1");
    const uint8_t* d_mask = nullptr;
    if (mask.has_value()) {
        CHECK_INPUT(*mask);
        TORCH_CHECK(mask->scalar_type() == torch::kUInt8 && mask->sizes() == depth.sizes(), "mask: uint8 (H,W)");
        d_mask = mask->data_ptr<uint8_t>();
    }
    egoprior_unproject_params p = {};
    p.H = (int32_t)depth.size(0);
    p.W = (int32_t)depth.size(1);
    p.subsample = (int32_t)subsample;
    p.select = (int32_t)select;
    p.fx = (float)intrinsics[0];
    p.fy = (float)intrinsics[1];
    p.cx = (float)intrinsics[2];
    p.cy = (float)intrinsics[3];
    copy_3x4<float>(c2w, p.c2w, "c2w");
    copy_3x4<double>(T_world, p.T, "T_world");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(depth.device());
    const int64_t hs = (p.H + subsample - 1) / subsample, ws_ = (p.W + subsample - 1) / subsample;
    const auto dev = depth.options();
    torch::Tensor points = torch::empty({hs * ws_, 4}, dev.dtype(torch::kFloat32));
    torch::Tensor count = torch::zeros({1}, dev.dtype(torch::kInt32));
    torch::Tensor flags = torch::zeros({1}, dev.dtype(torch::kInt32));
    torch::Tensor ws = torch::empty({(int64_t)egoprior_unproject_workspace_bytes(p.H, p.W, p.subsample)}, dev.dtype(torch::kUInt8));
    void* stream = current_stream(depth);
    int rc;
    {
        py::gil_scoped_release nogil;
        if (depth.scalar_type() == torch::kFloat16)
            rc = egoprior_unproject_frame_f16(&p, (const uint16_t*)depth.data_ptr(), rgb.data_ptr<uint8_t>(), d_mask,
                                              (egoprior_point*)points.data_ptr(), nullptr, count.data_ptr<int32_t>(),
                                              flags.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), stream);
        else
            rc = egoprior_unproject_frame(&p, depth.data_ptr<float>(), rgb.data_ptr<uint8_t>(), d_mask,
                                          (egoprior_point*)points.data_ptr(), nullptr, count.data_ptr<int32_t>(),
                                          flags.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), stream);
    }
    check(rc);
    return {points, count, flags};
}

// render_points_pytorch3d (:754-890) / render_points_fisheye (:892-1102) after their host-side set-up: clouds `a` (static)
// and optionally `b` (dynamic) -> (uint8 frame, status (1,) int32: 1 = the workspace was too short for the tile lists).
// camera = [R (9, row-major), T (3), fx, fy, px, py] (+ radial (6), tangential (2), thin prism (4) for model 1), all in
// pytorch3d's convention as include/egoprior.h documents for egoprior_camera.
std::vector<torch::Tensor> render_frame(int64_t model, int64_t H, int64_t W, std::vector<double> camera, double radius,
                                        double radius_sq_w, int64_t K, bool rotate_cw, int64_t colors_gt1,
                                        torch::Tensor a_points, torch::Tensor a_count, torch::Tensor a_flags,
                                        c10::optional<torch::Tensor> b_points, c10::optional<torch::Tensor> b_count,
                                        c10::optional<torch::Tensor> b_flags) {
    CHECK_INPUT(a_points);
    CHECK_INPUT(a_count);
    CHECK_INPUT(a_flags);
    TORCH_CHECK(a_points.dim() == 2 && a_points.size(1) == 4 && a_points.scalar_type() == torch::kFloat32,
                "points: (n, 4) float32 records");
    TORCH_CHECK(camera.size() == 16 || camera.size() == 28, "camera: 16 values (pinhole) or 28 (fisheye)");
    egoprior_camera cam = {};
    cam.model = (int32_t)model;
    cam.H = (int32_t)H;
    cam.W = (int32_t)W;
    for (int i = 0; i < 9; ++i) cam.R[i] = (float)camera[i];
    for (int i = 0; i < 3; ++i) cam.T[i] = (float)camera[9 + i];
    cam.fx = (float)camera[12];
    cam.fy = (float)camera[13];
    cam.px = (float)camera[14];
    cam.py = (float)camera[15];
    if (camera.size() == 28) {
        for (int i = 0; i < 6; ++i) cam.radial[i] = (float)camera[16 + i];
        for (int i = 0; i < 2; ++i) cam.tangential[i] = (float)camera[22 + i];
        for (int i = 0; i < 4; ++i) cam.thin_prism[i] = (float)camera[24 + i];
    }
    egoprior_raster_params rp = {};
    rp.radius = (float)radius;
    rp.radius_sq_w = (float)radius_sq_w;
    rp.K = (int32_t)K;
    rp.rotate_cw = rotate_cw ? 1 : 0;
    rp.colors_gt1 = (int32_t)colors_gt1;
    const egoprior_point* d_b = nullptr;
    const int32_t *d_nb = nullptr, *d_fb = nullptr;
    int64_t n_b = 0;
    if (b_points.has_value()) {
        TORCH_CHECK(b_count.has_value() && b_flags.has_value(), "the second cloud comes with its count and flags");
        CHECK_INPUT(*b_points);
        CHECK_INPUT(*b_count);
        CHECK_INPUT(*b_flags);
        TORCH_CHECK(b_points->dim() == 2 && b_points->size(1) == 4 && b_points->scalar_type() == torch::kFloat32,
                    "points: (n, 4) float32 records");
        d_b = (const egoprior_point*)b_points->data_ptr();
        d_nb = b_count->data_ptr<int32_t>();
        d_fb = b_flags->data_ptr<int32_t>();
        n_b = b_points->size(0);
    }
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(a_points.device());
    const int64_t n_a = a_points.size(0);
    const auto dev = a_points.options();
    torch::Tensor image = rotate_cw ? torch::empty({W, H, 3}, dev.dtype(torch::kUInt8)) : torch::empty({H, W, 3}, dev.dtype(torch::kUInt8));
    torch::Tensor status = torch::zeros({1}, dev.dtype(torch::kInt32));
    const size_t need = egoprior_render_workspace_bytes((int32_t)H, (int32_t)W, n_a + n_b, rp.radius);
    TORCH_CHECK(need != 0, "egoprior: bad raster size, point count or radius");
    torch::Tensor ws = torch::empty({(int64_t)need}, dev.dtype(torch::kUInt8));
    void* stream = current_stream(a_points);
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = egoprior_render_frame(&cam, &rp, (const egoprior_point*)a_points.data_ptr(), n_a, a_count.data_ptr<int32_t>(),
                                   a_flags.data_ptr<int32_t>(), d_b, n_b, d_nb, d_fb, image.data_ptr<uint8_t>(), nullptr,
                                   nullptr, nullptr, nullptr, status.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(),
                                   stream);
    }
    check(rc);
    return {image, status};
}

// vipe/utils/depth.py:385-422: (H, W) uint8 mask at full resolution
torch::Tensor reliable_depth_mask(torch::Tensor depth) {
    CHECK_INPUT(depth);
    TORCH_CHECK(depth.dim() == 2 && depth.scalar_type() == torch::kFloat32, "depth: (H,W) float32");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(depth.device());
    const int64_t H = depth.size(0), W = depth.size(1);
    torch::Tensor out = torch::empty({H, W}, depth.options().dtype(torch::kUInt8));
    void* stream = current_stream(depth);
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = egoprior_reliable_depth_mask(depth.data_ptr<float>(), (int32_t)H, (int32_t)W, out.data_ptr<uint8_t>(), stream);
    }
    check(rc);
    return out;
}

}  // namespace

void pybind_egoprior_ext(py::module& m) {
    m.def("unproject_frame", &unproject_frame, py::arg("depth"), py::arg("rgb"), py::arg("mask"), py::arg("intrinsics"),
          py::arg("c2w"), py::arg("T_world"), py::arg("subsample") = 2, py::arg("select") = (int64_t)EGOPRIOR_SELECT_STATIC);
    m.def("render_frame", &render_frame, py::arg("model"), py::arg("H"), py::arg("W"), py::arg("camera"), py::arg("radius"),
          py::arg("radius_sq_w"), py::arg("K"), py::arg("rotate_cw"), py::arg("colors_gt1"), py::arg("a_points"),
          py::arg("a_count"), py::arg("a_flags"), py::arg("b_points") = py::none(), py::arg("b_count") = py::none(),
          py::arg("b_flags") = py::none());
    m.def("reliable_depth_mask", &reliable_depth_mask, py::arg("depth"));
    m.attr("SELECT_ALL") = (int64_t)EGOPRIOR_SELECT_ALL;
    m.attr("SELECT_STATIC") = (int64_t)EGOPRIOR_SELECT_STATIC;
    m.attr("SELECT_DYNAMIC") = (int64_t)EGOPRIOR_SELECT_DYNAMIC;
    m.attr("CAM_PINHOLE") = (int64_t)EGOPRIOR_CAM_PINHOLE;
    m.attr("CAM_FISHEYE") = (int64_t)EGOPRIOR_CAM_FISHEYE;
}
