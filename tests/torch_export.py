"""Test fixture generator: an nn.Module of the NSNet2 architecture (fc1 -> GRU -> GRU -> relu(fc2) -> relu(fc3) ->
sigmoid(fc4), the graph of nsnet2-20ms-baseline.onnx that src/NSNet2.zig:53-112 binds) exported to ONNX by PyTorch's
own exporter -- node names, MatMul + Add instead of Gemm, the GRU's W / R / B layout and gate reordering exactly as a
real exporter writes them -- so that the library's ONNX reader is tested on a file it did not write itself.

The `onnx` Python package is not installed in this image.  torch's TorchScript exporter serialises the protobuf in
C++ and needs that package only for a post-processing hook that splices onnxscript custom functions into the
model (none here): the hook is bypassed, nothing else is touched.  (The dynamo exporter needs `onnxscript`, which
is absent too.)

Run as a script in a process of its own -- torch brings its own copy of the ROCm runtime, and once it is loaded
into a process the system librccl no longer finds the GPU ("no ROCm-capable device is detected"), so the test
session itself must not import torch:
    python tests/torch_export.py out.onnx out.npz n_fc1 n_hidden n_fc2 n_fc3 seed
writes the ONNX file and an .npz with the input `x` [1, 54, 161] and torch's own output `y`."""
import sys
import warnings

import numpy as np


def export_nsnet2(path, n_fc1=400, n_hidden=400, n_fc2=600, n_fc3=600, seed=0, T=54):
    """Writes `path`; returns (module, input [1, T, 161] float32, torch's own output [1, T, 161] float32)."""
    import torch
    import torch.nn as nn

    class NSNet2(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(161, n_fc1)
            self.rnn1 = nn.GRU(n_fc1, n_hidden, batch_first=True)
            self.rnn2 = nn.GRU(n_hidden, n_hidden, batch_first=True)
            self.fc2 = nn.Linear(n_hidden, n_fc2)
            self.fc3 = nn.Linear(n_fc2, n_fc3)
            self.fc4 = nn.Linear(n_fc3, 161)

        def forward(self, x):
            x = self.fc1(x)
            x, _ = self.rnn1(x)
            x, _ = self.rnn2(x)
            x = torch.relu(self.fc2(x))
            x = torch.relu(self.fc3(x))
            return torch.sigmoid(self.fc4(x))

    torch.manual_seed(seed)
    model = NSNet2().eval()
    with torch.no_grad():
        for p in model.parameters():          # larger than the default init: gates and gains leave their linear range
            p.mul_(2.5)
    x = (torch.rand(1, T, 161) * 13.0 - 11.0)
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as proto_utils
    proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model, (x,), path, input_names=["input"], output_names=["output"], dynamo=False, opset_version=13)
    with torch.no_grad():
        y = model(x)
    return model, x.numpy().astype(np.float32), y.numpy().astype(np.float32)


def export_in_subprocess(path, dims, seed=0):
    """export_nsnet2 in a child interpreter; returns (x, y) as numpy arrays"""
    import os
    import subprocess
    npz = path + ".npz"
    cmd = [sys.executable, os.path.abspath(__file__), path, npz] + [str(d) for d in dims] + [str(seed)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("torch ONNX export failed:\n" + r.stderr[-3000:])
    d = np.load(npz)
    return d["x"], d["y"]


if __name__ == "__main__":
    out_onnx, out_npz = sys.argv[1], sys.argv[2]
    f1, h, f2, f3, seed_ = (int(v) for v in sys.argv[3:8])
    _, x_, y_ = export_nsnet2(out_onnx, f1, h, f2, f3, seed=seed_)
    np.savez(out_npz, x=x_, y=y_)
