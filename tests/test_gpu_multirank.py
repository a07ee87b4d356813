"""bench.py's N > 1 path under ``pytest -m gpu``: two ranks started as FRESH child processes by
torch.distributed.run, sharing the one GPU of the box, collectives over gloo (SGNN_DIST_BACKEND=gloo: host memory) -- a
functional check of the data-parallel step, never a measurement.  What it pins until an 8-GPU node exists:
both scaling forms, both forms of the head, the pipelined schedule with its second communicator, the recorded
forward + backward of a strong shard -- and that the gradient exchange is exact: without dropout the loss after five
updates of the 2-rank strong runs is the single-rank loss (same subgraphs, draws keyed by global numbers)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
SIZE = ['--nodes', '100000', '--subgraphs', '2000', '--no-cpu-baseline', '--no-extras']


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _bench(ranks, extra, no_dropout=False):
    env = dict(os.environ, SGNN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    if no_dropout:
        env['SGNN_BENCH_HP'] = '{"lin_dropout": 0.0}'
    if ranks > 1:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), 'bench.py', '--gpus', str(ranks)]
    else:
        cmd = [sys.executable, 'bench.py', '--gpus', '1']
    r = subprocess.run(cmd + SIZE + extra, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (cmd + extra, r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)


@pytest.mark.parametrize('scaling,head', [('weak', 'sharded'), ('strong', 'replicated')])
def test_two_rank_bench_runs_pipelined(scaling, head):
    d = _bench(2, ['--steps', '2', '--warmup', '1', '--scaling', scaling, '--head', head, '--pipeline-multi'])
    assert d['n_gpus'] == 2 and d['scaling'] == scaling
    assert d['collectives']['rccl_ranks'] == 2 and d['collectives']['backend'] == 'gloo'
    assert d['config']['subgraphs_total'] == (4000 if scaling == 'weak' else 2000)
    assert d['config']['schedule']['passes_pipelined'] is True
    assert d['loss'] == d['loss'] and 0.5 < d['loss'] < 2.0                       # finite, a 3-class cross entropy near ln 3
    assert d['collectives']['north_star_exchange']['gathered_rows'] == 2 * d['collectives']['north_star_exchange']['rows_per_rank']
    assert d['collectives']['north_star_exchange']['inside_the_timed_step'] == (head == 'replicated')
    assert 'FUNCTIONAL CHECK' in d['data']


def test_two_rank_gradient_exchange_reproduces_the_single_rank_loss():
    """No dropout, 3 timed steps + 2 priming passes = 5 updates.  The same 2 000 subgraphs (a) on one rank, (b) dealt to two ranks
    with the head on the rank's own rows, forward + backward replayed from a hipGraph and the passes pipelined (second
    communicator), (c) the same eagerly and sequentially, (d) with the replicated head on the all-gathered embeddings: one
    loss.  (The clip + Adam arithmetic of the two paths differs in summation order only: 1e-5 relative.)"""
    common = ['--steps', '3', '--warmup', '0', '--scaling', 'strong']
    one = _bench(1, ['--steps', '3', '--warmup', '0'], no_dropout=True)
    graph = _bench(2, common + ['--head', 'sharded', '--pipeline-multi'], no_dropout=True)
    eager = _bench(2, common + ['--head', 'sharded', '--graph', 'off'], no_dropout=True)
    repl = _bench(2, common + ['--head', 'replicated'], no_dropout=True)
    assert graph['config']['schedule']['training_half_from_hipgraph'] is True and graph['config']['schedule']['passes_pipelined'] is True
    assert eager['config']['schedule']['training_half_from_hipgraph'] is False
    for name, d in (('recorded + pipelined', graph), ('eager', eager), ('replicated head', repl)):
        assert d['collectives']['rccl_ranks'] == 2 and d['config']['subgraphs_total'] == 2000, name
        assert abs(d['loss'] - one['loss']) <= 1e-5 * abs(one['loss']), (name, d['loss'], one['loss'])
    assert graph['loss'] == eager['loss']                          # a replayed step is the eager step, bit for bit


def test_every_schedule_of_the_single_rank_bench_trains_to_one_loss():
    """bench.py's schedules at N = 1 on one workload (2 000 subgraphs: above the 1 024 rows where the DTW stage decides
    about grouping repeated rows from counts that arrive asynchronously -- a recording of the preparation has to settle
    that before its capture starts): eager, training half from a hipGraph, both halves from hipGraphs, sequential.
    No dropout: one loss, bit for bit."""
    base = ['--steps', '4', '--warmup', '1']                       # 2 priming passes + 1 + 4 = 7 updates
    runs = {m: _bench(1, base + ['--graph', m], no_dropout=True) for m in ('off', 'train')}
    # (both halves recorded: bench.py primes with 4 passes there -- two eager ones, then one recording per slot: 4 + 1 + 2 = 7)
    runs['both'] = _bench(1, ['--steps', '2', '--warmup', '1', '--graph', 'both'], no_dropout=True)
    assert runs['both']['priming_passes_before_warmup'] == 4 and runs['off']['priming_passes_before_warmup'] == 2
    runs['sequential'] = _bench(1, base + ['--no-pipeline'], no_dropout=True)
    assert runs['both']['config']['schedule']['hipgraphs'] == 'both'
    assert runs['train']['config']['schedule']['training_half_from_hipgraph'] is True
    losses = {m: d['loss'] for m, d in runs.items()}
    assert len(set(losses.values())) == 1, losses
