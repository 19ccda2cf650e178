"""The file rendezvous of the native transport (include/blockcg_rccl.h: bcg_rccl_unique_id_via_file) is single-use, and
tools/launch_ranks.sh does not leave ranks waiting for a dead peer.  CPU-only: the test-only twin of the transport
(libblockcg_rccl_mock.so) creates ids without a GPU; the two-rank, same-idfile-twice run of the C++ driver is
tests/test_cpp_dropin.py."""
import ctypes
import os
import subprocess
import sys
import threading
import time

from conftest import ROOT

LAUNCH = os.path.join(ROOT, "tools", "launch_ranks.sh")


def _mock_lib():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "blockcg_amd", "csrc"), "-s", "mock"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = ctypes.CDLL(os.path.join(ROOT, "blockcg_amd", "_build", "libblockcg_rccl_mock.so"))
    lib.bcg_rccl_unique_id_via_file.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    return lib


def test_run_token_keeps_a_stale_id_file_from_being_read(tmp_path, monkeypatch):
    lib = _mock_lib()
    path = str(tmp_path / "bcg.id")
    with open(path, "wb") as f:           # what a launch that died before its communicator came up leaves behind
        f.write(b"S" * 128)
    monkeypatch.setenv("BCG_RUN_TOKEN", "launch-2")
    got = {}

    def reader():
        buf = ctypes.create_string_buffer(128)
        got["rc"] = lib.bcg_rccl_unique_id_via_file(path.encode(), 1, 20.0, buf)
        got["id"] = buf.raw

    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.3)                       # rank 1 is polling; without the token it would have returned the stale bytes
    assert t.is_alive()
    mine = ctypes.create_string_buffer(128)
    assert lib.bcg_rccl_unique_id_via_file(path.encode(), 0, 20.0, mine) == 0
    t.join(30)
    assert got["rc"] == 0 and got["id"] == mine.raw and got["id"] != b"S" * 128
    assert os.path.exists(path + ".launch-2")
    # and a reader of ANOTHER launch does not see this launch's file either
    monkeypatch.setenv("BCG_RUN_TOKEN", "launch-3")
    buf = ctypes.create_string_buffer(128)
    assert lib.bcg_rccl_unique_id_via_file(path.encode(), 1, 0.2, buf) != 0


def test_launch_ranks_sets_a_fresh_token_per_launch(tmp_path):
    out = [subprocess.run(["bash", LAUNCH, "2", "bash", "-c", "echo $RANK:$WORLD_SIZE:$BCG_RUN_TOKEN"], capture_output=True,
                          text=True, timeout=60) for _ in range(2)]
    toks = []
    for o in out:
        assert o.returncode == 0, o.stderr
        lines = sorted(o.stdout.split())
        assert [ln.split(":")[:2] for ln in lines] == [["0", "2"], ["1", "2"]]
        t = {ln.split(":")[2] for ln in lines}
        assert len(t) == 1 and t != {""}
        toks.append(t.pop())
    assert toks[0] != toks[1]


def test_launch_ranks_kills_the_survivors_of_a_failed_rank(tmp_path):
    marker = tmp_path / "pids"
    script = f'echo $$ >> {marker}; if [ "$RANK" = 1 ]; then exit 3; fi; sleep 600'
    t0 = time.time()
    r = subprocess.run(["bash", LAUNCH, "3", "bash", "-c", script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and time.time() - t0 < 30
    time.sleep(0.5)
    for pid in marker.read_text().split():
        assert not os.path.exists(f"/proc/{pid}"), pid
