"""bench.py --gpus N from a plain shell (no WORLD_SIZE): the parent must start its own ranks with
torch.distributed.run as a child process, without importing torch or touching the GPU itself, and
relay the child's output and return code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(kw)
    return env


def test_plain_shell_launch_builds_the_torchrun_command():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1"],
                         env=_env(BENCH_LAUNCH_DRY_RUN="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]      # the ranks see the same flags


def test_parent_does_not_import_torch_before_launching():
    probe = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2']\n"
             "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    code = e.code\n"
             "print('TORCH_IMPORTED' if 'torch' in sys.modules else 'TORCH_NOT_IMPORTED', code)\n" % BENCH)
    out = subprocess.run([sys.executable, "-c", probe], env=_env(BENCH_LAUNCH_DRY_RUN="1"),
                         capture_output=True, text=True, timeout=120)
    assert "TORCH_NOT_IMPORTED 0" in out.stdout, out.stdout + out.stderr


def test_child_return_code_is_relayed(tmp_path):
    """A launcher stand-in that fails: the parent exits with the child's code (the real launcher is
    torch.distributed.run; its own behaviour is covered on the GPU box by --rehearse-one-gpu)."""
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (fake / "__init__.py").write_text("")
    (fake / "run.py").write_text("import sys\nprint('{\"fake\": %d}' % len(sys.argv))\nsys.exit(7)\n")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(PYTHONPATH=str(tmp_path)),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 7
    assert '"fake"' in out.stdout


def _stub_env(**kw):
    return _env(BENCH_LAUNCH_MODULE="fake_launcher", PYTHONPATH=os.path.join(ROOT, "tests"), **kw)


def test_failed_run_on_the_library_communicator_is_repeated_on_the_torch_collective():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2"], env=_stub_env(),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["collective"] == "torch" and line["argv"][-2:] == ["--collective", "torch"]
    assert "--steps" in line["argv"] and "once more with --collective torch" in out.stderr


def test_stalled_run_on_the_library_communicator_is_stopped_and_repeated():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_stub_env(FAKE_LAUNCHER_STALL="1", BENCH_LIB_TIMEOUT="2"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1])["config"]["collective"] == "torch"
    assert "did not finish" in out.stderr


def test_an_explicit_collective_is_not_second_guessed():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--collective", "lib"], env=_stub_env(),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 3
