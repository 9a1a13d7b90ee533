// onnx_reader.h -- reads the ONNX model file the engine hands its executor.
//
// The reference passes an ONNX path to the executor's load()
// (/root/reference/src/infer/trt.cc:109-131; default "./res/model.onnx",
// src/context.h:93) and lets TensorRT's parser build whatever network the file holds, fixing
// only the tensor contract: input "input" [N,C,9,9], outputs "policy" (2187 per position),
// "value", "draw" (trt.cc:144-150,193-227).  This build runs ONE topology family on its own
// kernels, so the reader recognises exactly that family in a model
// file -- stem conv3x3 [+BatchNormalization | folded bias] + Relu, N x (conv-bn-relu-conv-bn-
// add-relu), 1x1 policy conv + Flatten/Reshape, 1x1 value conv [+BN] + Relu + Flatten + two
// dense layers (Gemm or MatMul+Add), value = (tanh + 1)/2 (Add/Mul/Div with constants) or
// Sigmoid, draw = Sigmoid -- and turns it into the NSGW v1 weight blob the loader consumes.
// Anything else is refused with a message naming the offending node.
//
// The protobuf wire format is decoded by hand from the public onnx.proto3 field numbers (no
// protobuf / onnx dependency).  Pinned against a model serialised by PyTorch's own exporter
// (tests/golden/net_torch_2x64.onnx, tests/test_onnx_io.py).
#ifndef NSG_ONNX_READER_H
#define NSG_ONNX_READER_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace nsg {
namespace onnx {

// True if the bytes start like an NSGW v1 weight file.
bool isNsgw(const void* data, size_t size);

// Converts an ONNX ModelProto into an NSGW v1 blob.  Returns false and sets *error otherwise.
bool convertToNsgw(const void* data, size_t size, std::vector<unsigned char>* blob, std::string* error);

} // namespace onnx
} // namespace nsg

#endif
