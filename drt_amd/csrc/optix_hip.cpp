// optix_hip.cpp -- boundary B1 as the reference binds it: a pybind11 class `optix_mesh` in a torch C++ extension.
//
// Drop-in for the reference's optix_extend.cpp (the same four members, optix_extend.cpp:77-83), loadable with the same
// call DiffRender.py:3-6 makes:
//
//     optix = torch.utils.cpp_extension.load(name="optix", sources=["<repo>/drt_amd/csrc/optix_hip.cpp"],
//         extra_include_paths=["<repo>/include", "/opt/rocm/include"], extra_cflags=["-D__HIP_PLATFORM_AMD__=1"],
//         extra_ldflags=["-L<repo>/drt_amd", "-ldrt_hip", "-Wl,-rpath,<repo>/drt_amd", "-lc10_hip"])
//     mesh = optix.optix_mesh(0); mesh.update_mesh(F, V); T, ID = mesh.intersect(Ray)
//
// (`python __graft_entry__.py` also builds it in-tree as drt_amd/optix*.so: `import drt_amd.optix as optix`.)
// Host-only C++: every member forwards to the C ABI of libdrt_hip.so (include/drt_hip.h) -- the LBVH build and the
// traversal kernels live there.  Differences from the reference class, all deliberate:
//   * inputs are validated with TORCH_CHECK (the reference has C asserts only and misreads non-contiguous tensors);
//   * work is enqueued on torch's CURRENT stream and nothing synchronises the host (OptiX Prime's execute(0) is
//     synchronous and ignores the stream); the GIL is released around the launches;
//   * `intersect` returns two owning, contiguous tensors (the reference returns strided aliases of one [N,2] buffer,
//     ID through a non-owning from_blob, optix_extend.cpp:49-55);
//   * the mesh is copied into the scene on update (the reference keeps raw device pointers into F and V alive through
//     member references, optix_extend.cpp:70-71).
// A miss has T = -1 and ID = -1; callers test T > 0 (reference DiffRender.py:391).
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <string>
#include <vector>

#include "drt_hip.h"

namespace {

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == DRT_OK, what, ": libdrt_hip error ", rc, ": ", drt_last_error());
}

torch::Tensor require(const torch::Tensor& t, c10::ScalarType dtype, int64_t cols, const char* name, int device) {
    TORCH_CHECK(t.defined(), name, " is undefined");
    TORCH_CHECK(t.scalar_type() == dtype, name, " must be ", c10::toString(dtype), ", got ", c10::toString(t.scalar_type()));
    TORCH_CHECK(t.dim() == 2 && t.size(1) == cols, name, " must have shape [N,", cols, "], got ", t.sizes());
    TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor (there is no CPU tracer in the product path)");
    TORCH_CHECK(t.get_device() == device, name, " is on device ", t.get_device(), ", this tracer is bound to device ", device);
    return t.contiguous();
}

}  // namespace

class optix_mesh {
public:
    explicit optix_mesh(unsigned cuda_device) : device_((int)cuda_device) {          // optix_extend.cpp:8-12
        check_rc(drt_create(device_, &scene_), "optix_mesh");
    }
    ~optix_mesh() {
        if (scene_) drt_destroy(scene_);
    }
    optix_mesh(const optix_mesh&) = delete;
    optix_mesh& operator=(const optix_mesh&) = delete;

    void update_mesh(torch::Tensor F, torch::Tensor V) {                              // optix_extend.cpp:14-21
        F = require(F, torch::kInt32, 3, "F", device_);
        V = require(V, torch::kFloat32, 3, "V", device_);
        const c10::DeviceGuard guard(V.device());
        void* stream = c10::hip::getCurrentHIPStream(device_).stream();
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = drt_update_mesh(scene_, F.data_ptr<int32_t>(), F.size(0), V.data_ptr<float>(), V.size(0), stream);
        }
        check_rc(rc, "update_mesh");
        builded_ = true;
    }

    void update_vert(torch::Tensor V) {                                               // optix_extend.cpp:23-27
        TORCH_CHECK(builded_, "update_mesh must be called first");
        V = require(V, torch::kFloat32, 3, "V", device_);
        const c10::DeviceGuard guard(V.device());
        void* stream = c10::hip::getCurrentHIPStream(device_).stream();
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = drt_update_vert(scene_, V.data_ptr<float>(), V.size(0), stream);
        }
        check_rc(rc, "update_vert");
    }

    std::vector<at::Tensor> intersect(torch::Tensor Ray) {                            // optix_extend.cpp:29-57
        TORCH_CHECK(builded_, "update_mesh must be called first");
        Ray = require(Ray, torch::kFloat32, 6, "Ray", device_);
        const c10::DeviceGuard guard(Ray.device());
        const int64_t n = Ray.size(0);
        auto T = torch::empty({n}, Ray.options());
        auto ID = torch::empty({n}, Ray.options().dtype(torch::kInt32));
        void* stream = c10::hip::getCurrentHIPStream(device_).stream();
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = drt_intersect(scene_, Ray.data_ptr<float>(), n, T.data_ptr<float>(), ID.data_ptr<int32_t>(), stream);
        }
        check_rc(rc, "intersect");
        return {T, ID};
    }

private:
    int device_;
    drt_scene_t* scene_ = nullptr;
    bool builded_ = false;
};

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "HIP LBVH tracer behind the reference's optix_mesh class (optix_extend.cpp)";
    py::class_<optix_mesh>(m, "optix_mesh", py::module_local())     // module_local: two builds of this file may live in one process
        .def(py::init<unsigned>())
        .def("update_mesh", &optix_mesh::update_mesh)
        .def("update_vert", &optix_mesh::update_vert)
        .def("intersect", &optix_mesh::intersect);
}
