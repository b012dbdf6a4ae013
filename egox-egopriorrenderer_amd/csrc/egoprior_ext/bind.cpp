// What the reference's csrc/bind.cpp gains for this module: a declaration and a sub-module, next to the seven it has
// (csrc/bind.cpp:20-49).  Built here as a module of its own so that the shim can be compiled and tested without the
// reference's other native code.
#include <torch/extension.h>

void pybind_egoprior_ext(py::module& m);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    py::module m_egoprior = m.def_submodule("egoprior_ext");
    pybind_egoprior_ext(m_egoprior);
}
