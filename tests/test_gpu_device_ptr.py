"""GPU: the zero-copy entry of the C ABI -- cl_chrom_create(on_device = 1) on torch tensors and a borrowed
torch stream (how a PyTorch host hands data over without a PCIe round trip).  Runs in a subprocess: torch has
to be imported BEFORE libcloops_hip.so so that both use the HIP runtime torch bundles."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
import torch
sys.path.insert(0, %r)
import numpy as np
import oracle
from cloops_amd import api
from cloops_amd.synth import synth_chrom
X, Y = synth_chrom(200000, 46709983, 13)
xt = torch.from_numpy(X).cuda()
yt = torch.from_numpy(Y).cuda()
stream = torch.cuda.Stream()
torch.cuda.synchronize()
ch = api.Chromosome.from_device_pointers(xt.data_ptr(), yt.data_ptr(), len(X), device=0, stream=stream.cuda_stream, keepalive=(xt, yt, stream))
for variant in ("v2", "v1", "block"):
    got = ch.cluster(variant, 2000, 5).labels
    want = oracle.labels(variant, X, Y, 2000, 5)
    assert np.array_equal(got, want), variant
a = ch.cluster("v2", 2000, 5, 3000).labels
b = api.Chromosome(X, Y).cluster("v2", 2000, 5, 3000).labels      # the copying constructor, private stream
assert np.array_equal(a, b)
ch.close()
assert torch.equal(xt.cpu(), torch.from_numpy(X)) and torch.equal(yt.cpu(), torch.from_numpy(Y))   # caller's arrays untouched
print("device pointer path ok")
'''


def test_device_pointers_and_borrowed_stream():
    pytest.importorskip("torch")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "device pointer path ok" in out.stdout
