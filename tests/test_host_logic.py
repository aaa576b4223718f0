"""Host-side logic that needs no GPU: Lasagne-like graph helpers, initialisers, parameter order,
gcnmain's data plumbing (dump.pkl format, geo_eval, haversine, flag parsing)."""
import gzip
import sys
import os
import pickle

import numpy as np
import pytest
import scipy.sparse as sps

from geographconv_amd import gcnmain
from geographconv_amd.nn import init, layers as L, nonlinearities as NL


def test_glorot_uniform_bounds_and_draw_order():
    np.random.seed(77)
    W = init.GlorotUniform(gain=1)((300, 200))
    a = np.sqrt(3) * np.sqrt(2.0 / 500)
    assert W.dtype == np.float32 and W.shape == (300, 200)
    assert np.abs(W).max() <= a and np.abs(W).max() > 0.98 * a
    np.random.seed(77)
    ref = np.random.uniform(-a, a, size=(300, 200)).astype(np.float32)     # Lasagne: one uniform draw
    assert np.array_equal(W, ref)
    assert np.all(init.Constant(-4.0)((7,)) == -4)
    with pytest.raises(RuntimeError):
        init.GlorotUniform()((5,))


def test_orthogonal_is_orthogonal():
    np.random.seed(1)
    Q = init.Orthogonal()((64, 64))
    assert np.allclose(Q.T @ Q, np.eye(64), atol=1e-5)
    R = init.Orthogonal()((32, 64))
    assert np.allclose(R @ R.T, np.eye(32), atol=1e-5)


def _graph(highway=True, hid=(8, 8, 8)):
    from geographconv_amd import gcnmodel as M
    np.random.seed(5)
    l_in = L.InputLayer((None, 20))
    l = M.SparseInputDenseLayer(l_in, num_units=hid[0], nonlinearity=NL.tanh)
    l = L.dropout(l, p=0.5)
    gates = []
    for h in hid[1:]:
        if highway:
            l, g = M.highway_dense(l, gconv=True, nonlinearity=NL.tanh, Wt=init.Orthogonal(), Wh=init.GlorotUniform(gain=1))
            gates.append(g)
        else:
            l = M.ConvolutionDenseLayer2(l, num_units=h, nonlinearity=NL.tanh)
    l_out = M.ConvolutionDenseLayer3(l, num_units=5, nonlinearity=NL.softmax)
    return l_in, l_out, gates


def test_param_order_matches_lasagne_get_all_params():
    l_in, l_out, gates = _graph(True)
    shapes = [p.shape for p in L.get_all_params(l_out)]
    # [W0,b0,(Wt,bt,Wh,bh)x2,Wo,bo]: gate before conv inside a block
    assert shapes == [(20, 8), (8,), (8, 8), (8,), (8, 8), (8,), (8, 8), (8,), (8, 8), (8,), (8, 5), (5,)]
    vals = L.get_all_param_values(l_out)
    assert np.all(vals[3] == -4.0) and np.all(vals[5] == 0.0)           # bt = -4, bh = 0
    assert np.allclose(vals[2].T @ vals[2], np.eye(8), atol=1e-5)       # Wt orthogonal, Wh not
    assert not np.allclose(vals[4].T @ vals[4], np.eye(8), atol=1e-2)
    # tags: biases are not regularizable
    assert len(L.get_all_params(l_out, regularizable=True)) == 6
    assert len(L.get_all_params(l_out, trainable=True)) == 12
    assert L.count_params(l_out) == 20 * 8 + 8 + 4 * (64 + 8) + 8 * 5 + 5
    l_in, l_out, _ = _graph(False, hid=(8, 6, 7))
    assert [p.shape for p in L.get_all_params(l_out)] == [(20, 8), (8,), (8, 6), (6,), (6, 7), (7,), (7, 5), (5,)]


def test_set_all_param_values_validates():
    l_in, l_out, _ = _graph(True)
    vals = L.get_all_param_values(l_out)
    vals[0] = vals[0] * 0 + 1
    L.set_all_param_values(l_out, vals)
    assert np.all(L.get_all_param_values(l_out)[0] == 1)
    with pytest.raises(ValueError):
        L.set_all_param_values(l_out, vals[:-1])
    vals[1] = np.zeros(9, np.float32)
    with pytest.raises(ValueError):
        L.set_all_param_values(l_out, vals)


def test_get_all_layers_topological_and_shapes():
    l_in, l_out, gates = _graph(True)
    layers = L.get_all_layers(l_out)
    assert layers[0] is l_in and layers[-1] is l_out
    pos = {l: i for i, l in enumerate(layers)}
    for l in layers:
        ins = getattr(l, 'input_layers', None) or ([l.input_layer] if getattr(l, 'input_layer', None) else [])
        assert all(pos[i] < pos[l] for i in ins)
    assert l_out.output_shape == (None, 5)
    assert gates[0].output_shape == (None, 8)


def test_gating_layer_asserts_equal_shapes():
    from geographconv_amd import gcnmodel as M
    a = L.InputLayer((None, 4))
    b = L.InputLayer((None, 5))
    with pytest.raises(AssertionError):
        M.MultiplicativeGatingLayer(a, a, b)


def test_haversine_known_distances():
    assert gcnmain.haversine((0, 0), (0, 0)) == 0
    # one degree of longitude on the equator
    assert abs(gcnmain.haversine((0, 0), (0, 1)) - 111.195) < 0.01
    # Lyon - Paris, the example of the `haversine` package the reference imports
    assert abs(gcnmain.haversine((45.7597, 4.8422), (48.8567, 2.3508)) - 392.217) < 0.05


def test_geo_eval_metrics():
    users = ['a', 'b', 'c']
    loc = {'a': '0,0', 'b': '0,0', 'c': '10,10'}
    lat = {'0': 0.0, '1': 0.0, '2': 50.0}
    lon = {'0': 0.0, '1': 1.0, '2': 50.0}
    mean, median, acc, dist, t, p = gcnmain.geo_eval(None, np.array([0, 1, 2]), users, lat, lon, loc)
    assert dist[0] == 0 and abs(dist[1] - 111.195) < 0.01 and dist[2] > 161
    assert abs(acc - 200.0 / 3) < 1e-9 and median == dist[1]
    with pytest.raises(AssertionError):
        gcnmain.geo_eval(None, np.array([0]), users, lat, lon, loc)


def test_dump_pkl_roundtrip_and_preprocess(tmp_path, monkeypatch):
    d = tmp_path / 'data'
    d.mkdir()
    A = sps.identity(4, format='csr', dtype=np.float32)
    X = sps.random(4, 6, density=0.5, format='csr', dtype=np.float32, random_state=0)
    Y = np.array([0, 1, 0, 1])
    data = (A, X[:2], Y[:2], X[2:3], Y[2:3], X[3:], Y[3:], ['u0', 'u1'], ['u2'], ['u3'], {'0': 1.0, '1': 2.0},
            {'0': 3.0, '1': 4.0}, {'u%d' % i: '1,1' for i in range(4)})
    gcnmain.dump_obj(data, str(d / 'dump.pkl'))
    with gzip.open(str(d / 'dump.pkl'), 'rb') as f:          # the reference's format: gzip + pickle (data.py:28-34)
        assert len(pickle.load(f)) == 13
    args = gcnmain.parse_args(['-d', str(d), '-hid', '300', '300', '300', '-highway', '-dropout', '0.5', '-reg', '0.0'])
    monkeypatch.setattr(gcnmain, 'model_args', args)
    back = gcnmain.preprocess_data(str(d))
    assert len(back) == 13 and (back[0] != A).nnz == 0 and back[7] == ['u0', 'u1']
    # no dump and no --synthetic: a clear error, not a silent fallback
    args2 = gcnmain.parse_args(['-d', str(tmp_path / 'nope')])
    monkeypatch.setattr(gcnmain, 'model_args', args2)
    with pytest.raises(FileNotFoundError):
        gcnmain.preprocess_data(str(tmp_path / 'nope'))


def test_dump_pkl_written_by_python2_is_read(tmp_path):
    """The dump.pkl a user of the reference holds is a Python-2 pickle (reference README.md:32; gcnmain.py:85-98 writes the 13-tuple):
    8-bit strings -- the raw bytes of every array among them --, `scipy.sparse.csr.csr_matrix`, `numpy.core.multiarray`.  load_obj must
    read it without help (plain pickle.load raises UnicodeDecodeError on it) and without leaning on scipy's deprecated module aliases."""
    import pickletools
    import warnings
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from py2_pickle import dumps
    A = sps.random(50, 50, density=0.1, format='csr', dtype=np.float32, random_state=1)
    X = sps.random(50, 30, density=0.2, format='csr', dtype=np.float32, random_state=0)
    Y = np.arange(50) % 5
    users = ['user%d' % i for i in range(49)] + [u'us\u00e9r49']          # one non-ASCII name: a Python-2 `unicode`
    data = (A, X[:30], Y[:30], X[30:40], Y[30:40], X[40:], Y[40:], users[:30], users[30:40], users[40:],
            {str(c): 30.0 + c for c in range(5)}, {str(c): -100.0 + c for c in range(5)}, {u: '40.7,-74.0' for u in users})
    raw = dumps(data)
    ops = {o.name for o, _, _ in pickletools.genops(raw)}
    assert 'SHORT_BINSTRING' in ops and 'BINSTRING' in ops and 'BINBYTES' not in ops and 'SHORT_BINBYTES' not in ops
    globs = {a for o, a, _ in pickletools.genops(raw) if o.name == 'GLOBAL'}
    assert 'scipy.sparse.csr csr_matrix' in globs and 'numpy.core.multiarray _reconstruct' in globs
    with pytest.raises(UnicodeDecodeError):
        pickle.loads(raw)
    f = tmp_path / 'dump.pkl'
    with gzip.open(str(f), 'wb') as fout:
        fout.write(raw)
    with warnings.catch_warnings():
        warnings.simplefilter('error')          # (scipy's "namespace is deprecated" warning would be an error here)
        back = gcnmain.load_obj(str(f))
    assert len(back) == 13 and sps.isspmatrix_csr(back[0]) and back[0].dtype == np.float32
    assert (back[0] != A).nnz == 0 and (back[5] != X[40:]).nnz == 0 and np.array_equal(back[2], Y[:30]) and back[2].dtype == Y.dtype
    assert back[7] == users[:30] and back[9][-1] == u'us\u00e9r49' and all(isinstance(u, str) for u in back[9])
    assert back[10] == data[10] and back[12] == data[12]


def test_reference_flag_set_is_accepted():
    # README.md:167,173 commands of the reference
    a = gcnmain.parse_args('-hid 300 300 300 -bucket 50 -batch 500 -d ./data/cmu -mindf 10 -reg 0.0 -dropout 0.5 -cel 5 -highway'.split())
    assert a.hid == [300, 300, 300] and a.highway and a.dropout == 0.5 and a.regularization == 0.0 and a.bucket == 50
    a = gcnmain.parse_args('-hid 600 600 600 -bucket 2400 -batch 500 -d ./data/na -mindf 10 -reg 0.0 -dropout 0.5 -cel 15 -highway'.split())
    assert a.hid == [600, 600, 600] and a.celebrity == 15
    a = gcnmain.parse_args([])
    assert a.hid == [100] and a.seed == 77 and a.lblfraction == [1.0] and a.maxdown == 10 and not a.highway


def test_synthetic_data_tuple_shape():
    data = gcnmain.synthetic_data('cmu')
    A, Xtr, Ytr, Xdv, Ydv, Xte, Yte, Utr, Udv, Ute, clat, clon, uloc = data
    assert A.shape == (9475, 9475) and Xtr.shape[0] + Xdv.shape[0] + Xte.shape[0] == 9475
    assert len(Utr) == Xtr.shape[0] and set(clat) == set(str(c) for c in range(129))
    assert all(u in uloc for u in Ute[:5])


def test_content_keys_are_128_bit_and_frozen_arrays_are_hashed_once():
    """Device copies of index / label vectors are keyed by CONTENT (128-bit hash): an in-place edit changes the key; a
    read-only array that owns its data is hashed once per object; a read-only VIEW of a writable array is not trusted."""
    from geographconv_amd import gcnmodel as M
    a = np.arange(1000, dtype=np.int32)
    k1 = M._content_key(a)
    assert max(M._content_key(np.arange(n))[2].bit_length() for n in range(1, 9)) > 64           # a 128-bit digest
    a[3] = 7
    k2 = M._content_key(a)
    assert k1 != k2 and M._content_key(a.copy()) == k2
    a.setflags(write=False)
    calls = []
    real = M._hash_bytes
    M._hash_bytes = lambda mv: calls.append(1) or real(mv)
    try:
        assert M._content_key(a) == k2 and M._content_key(a) == k2 and M._content_key(a) == k2
        assert len(calls) == 1                                  # hashed once, then remembered for this object
        b = np.arange(1000, dtype=np.int32)
        v = b[:500]
        v.setflags(write=False)
        M._content_key(v)
        M._content_key(v)
        assert len(calls) == 3                                  # a view of a writable base: rehashed every time
    finally:
        M._hash_bytes = real
    assert M._content_key(None) is None


def test_product_code_reads_only_the_documented_environment_variables():
    """geographconv_amd/tuning.py is the one table: four environment variables (+ the build script's GEOGCN_BUILD_DEFINES), no
    getenv() in the kernels outside an ablation build -- except through the ONE test-seam reader (csrc/core.hip test_seam_i64), whose two
    names are pinned here as well."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'geographconv_amd')
    seen, seams = set(), set()
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                if f.endswith('.py'):
                    seen.update(re.findall(r"environ(?:\.get)?\(?\[?\s*['\"](GEOGCN_[A-Z0-9_]+)", src))
                else:
                    src = re.sub(r'#ifdef GEOGCN_BF16_PROBE_BUILD.*?#endif', '', src, flags=re.S)
                    seen.update(re.findall(r'getenv\("(GEOGCN_[A-Z0-9_]+)"', src))
                    assert len(re.findall(r'\bgetenv\(', src)) == (1 if f == 'core.hip' else 0), f      # only test_seam_i64 calls getenv
                    seams.update(re.findall(r'test_seam_i64\("(GEOGCN_[A-Z0-9_]+)"', src))
    assert seams == {'GEOGCN_X3_ROWS_MIN_M', 'GEOGCN_TN_SLAB_LIMIT'}, seams
    assert seen == {'GEOGCN_GEMM_PRECISION', 'GEOGCN_HIP_GRAPH', 'GEOGCN_DIST_EXCHANGE', 'GEOGCN_DIST_BACKEND',
                    'GEOGCN_BUILD_DEFINES'}, seen


def test_bench_starts_its_own_ranks_when_no_launcher_did(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset re-executes itself under torch.distributed.run: one process per GPU on
    127.0.0.1 and a free port, the caller's flags handed through, the ranks' exit code returned."""
    import importlib.util
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        seen.setdefault('ports', []).append(cmd[cmd.index('--master-port') + 1])
        return R()
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1', '--shape', 'cmu'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 1024 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    assert cmd[-8:] == ['--gpus', '4', '--steps', '3', '--warmup', '1', '--shape', 'cmu'] and cmd[-9].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert len(seen['ports']) == 2          # a launch that dies at once (here: code 7 in no time) is tried once more, on a port found anew


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_live_traffic_reads_the_counter_tables(tmp_path, monkeypatch):
    """bench.py --traffic live: the two rocprofv3 --pmc child passes are parsed per dispatch (rows of one dispatch summed), only the named
    kernel counts, 2 x FETCH_SIZE + WRITE_SIZE kilobytes -> bytes; a pass without the kernel, a missing tool and a profiler around the
    invocation itself are reported as errors (the line then quotes the stored pass and says so)."""
    import subprocess
    bench = _load_bench()
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size",'
            '"LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n')
    kern = 'void geogcn::(anonymous namespace)::spmm_rows_kernel<5, 0, 16, 0, 0>(geogcn::SpmmArgs)'

    def table(counter, values, other=123.0):
        rows = [head]
        for i, parts in enumerate(values):
            for v in parts:          # (a dispatch split over several rows)
                rows.append('%d,%d,"Agent 2",1,1,1,64,3,"%s",256,0,0,48,0,64,"%s",%f,10,20\n' % (i + 1, i + 1, kern, counter, v))
        rows.append('99,99,"Agent 2",1,1,1,64,4,"void geogcn::(anonymous namespace)::spmm_rows_kernel<5, 1, 16, 0, 1>(x)",256,0,0,48,0,64,"%s",%f,10,20\n' % (counter, other))
        return ''.join(rows)

    calls = []

    def fake_run(cmd, cwd=None, env=None, **kw):
        counter = cmd[cmd.index('--pmc') + 1]
        d = cmd[cmd.index('-d') + 1]
        os.makedirs(os.path.join(d, 'host'), exist_ok=True)
        vals = {'FETCH_SIZE': [[1000.0, 1000.0], [3000.0]], 'WRITE_SIZE': [[500.0], [700.0]]}[counter]
        if not (counter == 'WRITE_SIZE' and calls and calls[0] == 'drop'):
            open(os.path.join(d, 'host', 'p_counter_collection.csv'), 'w').write(table(counter, vals))
        calls.append(counter)
        assert '--pmc-child' in cmd and cwd == '/tmp' and env['TMPDIR'] == '/tmp'

        class R:
            returncode, stderr = 0, ''
        return R()
    for k in list(os.environ):
        if k.startswith(('ROCPROF', 'ROCP_')):
            monkeypatch.delenv(k)
    monkeypatch.setenv('LD_PRELOAD', '')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(bench.os.path, 'exists', lambda p, _e=os.path.exists: True if p.endswith('rocprofv3') else _e(p))
    r = bench.traffic_live(['--pmc-child'], 'spmm_rows_kernel<5, 0, 16, 0, 0>')
    assert r['fetch_size_kb'] == 2500.0 and r['write_size_kb'] == 600.0 and r['launches_counted'] == {'FETCH_SIZE': 2, 'WRITE_SIZE': 2}
    assert r['bytes'] == (2 * 2500.0 + 600.0) * 1024
    calls[:] = ['drop']
    r = bench.traffic_live(['--pmc-child'], 'spmm_rows_kernel<5, 0, 16, 0, 0>')
    assert 'error' in r and 'WRITE_SIZE' in r['error']
    monkeypatch.setenv('ROCPROFILER_REGISTER_FORCE_LOAD', '1')
    assert 'profiler' in bench.traffic_live(['--pmc-child'], 'x')['error']


def test_dense_head_size_follows_the_cost_model():
    """ops.dense_head_size: the head panel of X^T . G takes whole GEMM tiles of the densest columns while a padded row of the
    split-K product (2 N flop per output column) is cheaper than gathering the entries it removes from the tail."""
    from geographconv_amd import ops, synth
    s = synth.SHAPES['twus']
    X = synth.bow_x(s.N, s.V, s.mean_nnz)
    col = np.diff(sps.csc_matrix(X).indptr)
    assert ops.dense_head_size(col, s.N) == 256                        # (the 3.5 % density rule took 180 columns = 256 padded rows)
    assert ops.dense_head_size(np.full(5000, 40), 100000) == 0         # uniformly sparse columns: no head
    assert ops.dense_head_size(np.r_[np.full(160, 60000), np.full(5000, 40)], 100000) == 160
    assert ops.dense_head_size(np.r_[np.full(100, 60000), np.full(20, 40)], 100000) == 0     # fewer columns than a tile
    assert ops.dense_head_size(np.zeros(10, dtype=np.int64), 50) == 0
