"""HIP runtime load order (blockcg_amd/_lib.py: single_hip_runtime): importing blockcg_amd BEFORE torch must leave the process
with one ROCr instance -- torch's bundled one, mapped without importing torch -- so that a later `import torch` still
sees the GPU; BCG_HIP_RUNTIME=system opts out; a foreign, already mapped system runtime is reported, not ignored."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(code, **env):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-W", "always", "-c", f"import sys; sys.path.insert(0, {ROOT!r})\n" + code], env=e,
                          capture_output=True, text=True, timeout=300)


def test_library_first_maps_torchs_bundled_runtime_without_importing_torch():
    r = _run("import blockcg_amd\n"
             "blockcg_amd.load()\n"
             "from blockcg_amd import _lib\n"
             "assert 'torch' not in sys.modules\n"
             "maps = open('/proc/self/maps').read()\n"
             "assert _lib.HIP_RUNTIME == 'torch-bundle', _lib.HIP_RUNTIME\n"
             "assert 'torch/lib/libamdhip64.so' in maps and 'torch/lib/libhsa-runtime64.so' in maps\n"
             "import torch\n"
             "n = sum(1 for ln in open('/proc/self/maps') if 'torch/lib/libamdhip64.so' in ln and ' r-xp ' in ln)\n"
             "assert n == 1, n\n"
             "print('ORDER_OK')\n")
    assert r.returncode == 0 and "ORDER_OK" in r.stdout, r.stdout + r.stderr


def test_torch_first_is_left_alone_and_opt_out_works():
    r = _run("import torch, blockcg_amd\nblockcg_amd.load()\nfrom blockcg_amd import _lib\nassert _lib.HIP_RUNTIME == 'torch'\nprint('OK1')\n")
    assert r.returncode == 0 and "OK1" in r.stdout, r.stdout + r.stderr
    r = _run("import blockcg_amd\nblockcg_amd.load()\nfrom blockcg_amd import _lib\nassert _lib.HIP_RUNTIME == 'system'\n"
             "assert 'torch/lib/libamdhip64.so' not in open('/proc/self/maps').read()\nprint('OK2')\n", BCG_HIP_RUNTIME="system")
    assert r.returncode == 0 and "OK2" in r.stdout, r.stdout + r.stderr


def test_foreign_system_runtime_already_mapped_is_reported():
    r = _run("import ctypes\nctypes.CDLL('/opt/rocm/lib/libamdhip64.so.7', mode=ctypes.RTLD_GLOBAL)\n"
             "import blockcg_amd\nblockcg_amd.load()\nfrom blockcg_amd import _lib\nassert _lib.HIP_RUNTIME == 'system'\nprint('OK3')\n")
    assert r.returncode == 0 and "OK3" in r.stdout, r.stdout + r.stderr
    assert "Import torch before" in r.stderr and "RuntimeWarning" in r.stderr, r.stderr


@pytest.mark.gpu
def test_library_first_then_torch_both_see_the_gpu():
    r = _run("import blockcg_amd as bc\n"
             "ctx = bc.Context([8, 4, 4, 4])\n"
             "x = bc.block_fermion_field(ctx, 4).setRandom(seed=1)\n"
             "import torch\n"
             "assert torch.cuda.is_available() and torch.cuda.device_count() >= 1\n"
             "t = torch.ones(1024, device='cuda').sum().item()\n"
             "assert t == 1024.0\n"
             "assert abs(x.download()).max() > 0\n"
             "print('BOTH_OK')\n")
    assert r.returncode == 0 and "BOTH_OK" in r.stdout, r.stdout + r.stderr
