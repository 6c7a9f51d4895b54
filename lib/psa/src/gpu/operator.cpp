// pybind module `psamask_gpu` (built by lib/psa/src/__init__.py with torch.utils.cpp_extension): at::Tensor
// front-end of semseg_psamask_forward / semseg_psamask_backward.  Contract kept from the reference
// (lib/psa/functions/psamask.py:17,31): the CALLER allocates and zero-fills the destination, the callee writes the
// in-window elements in place and returns nothing.  Unlike the reference (no checks, `.data<float>()` on whatever
// it is handed, legacy default stream) this front-end validates device / dtype / layout / shape and launches on
// the caller's current stream; a failing C-ABI call raises.
#include "operator.h"

#include <c10/hip/HIPStream.h>

#include "semseg_hip.h"

namespace {

void check_pair(const at::Tensor& src, const at::Tensor& dst, const char* what, int64_t n, int64_t src_c,
                int64_t dst_c, int64_t h, int64_t w) {
  TORCH_CHECK(src.is_cuda() && dst.is_cuda(), what, ": tensors must live on the MI355X (no CPU path in psamask_gpu)");
  TORCH_CHECK(src.get_device() == dst.get_device(), what, ": source and destination on different devices");
  TORCH_CHECK(src.scalar_type() == at::kFloat && dst.scalar_type() == at::kFloat, what, ": float32 only");
  TORCH_CHECK(src.is_contiguous() && dst.is_contiguous(), what, ": dense NCHW tensors expected");
  TORCH_CHECK(src.dim() == 4 && src.size(0) == n && src.size(1) == src_c && src.size(2) == h && src.size(3) == w,
              what, ": source is not [", n, ",", src_c, ",", h, ",", w, "]");
  TORCH_CHECK(dst.dim() == 4 && dst.size(0) == n && dst.size(1) == dst_c && dst.size(2) == h && dst.size(3) == w,
              what, ": destination is not [", n, ",", dst_c, ",", h, ",", w, "]");
}

hipStream_t stream_of(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

}  // namespace

void psamask_forward_cuda(const int psa_type, const at::Tensor& input, at::Tensor& output, const int num_,
                          const int feature_H_, const int feature_W_, const int mask_H_, const int mask_W_,
                          const int half_mask_H_, const int half_mask_W_) {
  check_pair(input, output, "psamask_forward", num_, (int64_t)mask_H_ * mask_W_, (int64_t)feature_H_ * feature_W_,
             feature_H_, feature_W_);
  const int rc = semseg_psamask_forward(psa_type, input.data_ptr<float>(), output.data_ptr<float>(), num_,
                                        feature_H_, feature_W_, mask_H_, mask_W_, half_mask_H_, half_mask_W_,
                                        stream_of(input));
  TORCH_CHECK(rc == 0, "semseg_psamask_forward failed with code ", rc);
}

void psamask_backward_cuda(const int psa_type, const at::Tensor& grad_output, at::Tensor& grad_input, const int num_,
                           const int feature_H_, const int feature_W_, const int mask_H_, const int mask_W_,
                           const int half_mask_H_, const int half_mask_W_) {
  check_pair(grad_output, grad_input, "psamask_backward", num_, (int64_t)feature_H_ * feature_W_,
             (int64_t)mask_H_ * mask_W_, feature_H_, feature_W_);
  const int rc = semseg_psamask_backward(psa_type, grad_output.data_ptr<float>(), grad_input.data_ptr<float>(), num_,
                                         feature_H_, feature_W_, mask_H_, mask_W_, half_mask_H_, half_mask_W_,
                                         stream_of(grad_output));
  TORCH_CHECK(rc == 0, "semseg_psamask_backward failed with code ", rc);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("psamask_forward", &psamask_forward_cuda, "PSAMASK forward (gfx950, libsemseg_hip.so)");
  m.def("psamask_backward", &psamask_backward_cuda, "PSAMASK backward (gfx950, libsemseg_hip.so)");
}
