// Native operator interface of lib/psa on the MI355X build.  Same two entry points, argument order and
// semantics as the reference's lib/psa/src/gpu/operator.h:3-4 (the pybind module is named psamask_gpu and
// exports psamask_forward / psamask_backward, lib/psa/src/gpu/operator.cpp:3-6), so the reference's
// lib/psa/functions/psamask.py:18-22,32-35 calls it unchanged.  The bodies forward to the C ABI of
// libsemseg_hip.so (include/semseg_hip.h) on torch's CURRENT HIP stream.
#pragma once
#include <torch/extension.h>

void psamask_forward_cuda(const int psa_type, const at::Tensor& input, at::Tensor& output, const int num_,
                          const int feature_H_, const int feature_W_, const int mask_H_, const int mask_W_,
                          const int half_mask_H_, const int half_mask_W_);
void psamask_backward_cuda(const int psa_type, const at::Tensor& grad_output, at::Tensor& grad_input, const int num_,
                           const int feature_H_, const int feature_W_, const int mask_H_, const int mask_W_,
                           const int half_mask_H_, const int half_mask_W_);
