"""CPU: `python3 bench.py --gpus N` without torch.distributed.run in the environment must turn
itself into a launcher -- a parent that never touches the GPU and starts the ranks as a fresh
child process -- and relay rank 0's JSON line and the child's exit code."""
import json
import os
import subprocess
import sys
import types

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, 'bench.py')


def _import_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_child_command_line():
    bench = _import_bench()
    cmd = bench.self_launch_command(4, ['--gpus', '4', '--steps', '20', '--warmup', '5'], 29512)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29512'
    at = cmd.index(BENCH)
    assert cmd[at + 1:] == ['--gpus', '4', '--steps', '20', '--warmup', '5']


@pytest.mark.parametrize('rc, stdout, expect_rc, expect_line', [
    (0, 'noise\n{"metric": "m", "value": 1.0, "n_gpus": 2}\ntrailing\n', 0, True),
    (3, 'rank 1 died\n', 3, False),
    (0, 'no line at all\n', 1, False),                 # rc 0 without a result is still a failure
    (7, '{"metric": "m", "value": 2.0}\n', 7, True),   # the line is relayed, the code kept
])
def test_relay_of_line_and_exit_code(capsys, rc, stdout, expect_rc, expect_line):
    bench = _import_bench()
    seen = {}

    def fake_run(cmd, env, stdout=None, text=None, _out=stdout):
        seen['cmd'], seen['env'] = cmd, env
        return types.SimpleNamespace(returncode=rc, stdout=_out)

    mods = [m for m in bench.GPU_MODULES if m in sys.modules]
    saved = {m: sys.modules.pop(m) for m in mods}      # (this test process has torch imported)
    try:
        got = bench.self_launch(2, ['--gpus', '2'], run=fake_run)
    finally:
        sys.modules.update(saved)
    out = capsys.readouterr().out
    assert got == expect_rc
    assert seen['env']['FFK_BENCH_LAUNCHER'] == 'self'
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert not any(k in seen['env'] for k in ('RANK', 'LOCAL_RANK'))
    lines = [ln for ln in out.splitlines() if ln.strip()]
    if expect_line:
        assert len(lines) == 1 and json.loads(lines[0])['metric'] == 'm'
    else:
        assert lines == []


def test_launcher_refuses_to_run_from_a_process_that_loaded_gpu_modules(monkeypatch):
    bench = _import_bench()
    if 'torch' not in sys.modules:
        monkeypatch.setitem(sys.modules, 'torch', types.ModuleType('torch'))
    with pytest.raises(RuntimeError, match='must stay off the GPU'):
        bench.self_launch(2, ['--gpus', '2'], run=lambda *a, **k: None)


def test_parent_process_end_to_end_stays_off_the_gpu(tmp_path):
    """The real thing in a real process: bench.py --gpus 2 with a stand-in for
    `python -m torch.distributed.run` first on PYTHONPATH.  The stand-in records what it was given
    and plays rank 0; the parent must exit with its code, print exactly its line, and -- checked
    through a sitecustomize hook at interpreter exit -- never have imported torch or the package."""
    fake = tmp_path / 'torch' / 'distributed'
    fake.mkdir(parents=True)
    # the parent's environment lacks FFK_BENCH_LAUNCHER, the child's has it (set by the launcher)
    (tmp_path / 'torch' / '__init__.py').write_text(
        "import os\nif os.environ.get('FFK_BENCH_LAUNCHER') != 'self':\n"
        "    open(os.environ['FFK_TEST_LOG'], 'a').write('PARENT IMPORTED TORCH\\n')\n")
    (fake / '__init__.py').write_text('')
    (fake / 'run.py').write_text(
        "import json, os, sys\n"
        "open(os.environ['FFK_TEST_LOG'], 'a').write('child argv ' + json.dumps(sys.argv[1:]) + '\\n')\n"
        "print('some rank chatter')\n"
        "print(json.dumps({'metric': 'x', 'n_gpus': 2, 'launcher': os.environ.get('FFK_BENCH_LAUNCHER')}))\n"
        "sys.exit(5)\n")
    (tmp_path / 'sitecustomize.py').write_text(
        "import atexit, os, sys\n"
        "def _report():\n"
        "    if os.environ.get('FFK_BENCH_LAUNCHER') == 'self':\n"
        "        return\n"
        "    bad = [m for m in ('torch', 'filter_functions_amd', 'filter_functions_amd._lib') if m in sys.modules]\n"
        "    open(os.environ['FFK_TEST_LOG'], 'a').write('parent modules ' + repr(bad) + '\\n')\n"
        "atexit.register(_report)\n")
    log = tmp_path / 'log.txt'
    env = dict(os.environ, PYTHONPATH=str(tmp_path), FFK_TEST_LOG=str(log))
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'FFK_BENCH_LAUNCHER'):
        env.pop(key, None)
    res = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '20', '--warmup', '5'],
                         env=env, capture_output=True, text=True, timeout=120)
    text = log.read_text()
    assert res.returncode == 5, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {'metric': 'x', 'n_gpus': 2, 'launcher': 'self'}
    assert 'PARENT IMPORTED TORCH' not in text
    assert "parent modules []" in text
    child = [ln for ln in text.splitlines() if ln.startswith('child argv')][0]
    argv = json.loads(child[len('child argv '):])
    assert '--nproc-per-node=2' in argv and argv[-6:] == ['--gpus', '2', '--steps', '20', '--warmup', '5']
