"""The reference's entry point end to end on the GPU: `gcnmain.py` flags -> data tuple -> GraphConv ->
fit -> predict -> geo_eval, on the pinned CMU-shape synthetic data."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gcnmain_runs_reference_flag_set(tmp_path, monkeypatch):
    from geographconv_amd import gcnmain
    monkeypatch.chdir(tmp_path)
    argv = '-hid 300 300 300 -bucket 50 -batch 500 -d ./data/cmu -mindf 10 -reg 0.0 -dropout 0.5 -cel 5 -highway -silent --synthetic cmu --epochs 12 -maxdown 3 -save'.split()
    clf, results = gcnmain.run(argv)
    assert clf.fitted and len(results) == 1
    dev_mean, dev_median, dev_acc = results[0]['dev']
    assert np.isfinite(dev_mean) and 0 <= dev_acc <= 100
    assert os.path.exists('gcn_1.0_percent_pred_129.pkl')            # gcnmain.py:228
    assert os.path.exists('./data/model-9475-1.0.pkl')               # gcnmain.py:196,223
    # -load path restores the same predictions
    clf2, _ = gcnmain.run(argv[:-1] + ['-load'])
    data = gcnmain.synthetic_data('cmu')
    import scipy.sparse as sps
    X = sps.vstack([data[1], data[3], data[5]]).tocsr().astype('float32')
    idx = np.arange(100, dtype=np.int32)
    p1, _ = clf.predict(X, data[0], idx)
    p2, _ = clf2.predict(X, data[0], idx)
    assert np.array_equal(p1, p2)


def test_feature_report_probes_model_with_onehot_inputs(tmp_path, monkeypatch):
    """-feature_report (reference gcnmain.py:234-261): predict on X = I (vocabulary) with A = I -- a second
    (X, A) pair of a different node count through the same trained model."""
    from geographconv_amd import gcnmain
    monkeypatch.chdir(tmp_path)
    argv = '-hid 64 64 -reg 0.0 -dropout 0.0 -highway -silent --synthetic cmu --epochs 3 -maxdown 1 -feature_report'.split()
    clf, results = gcnmain.run(argv)
    txt = open('important_features.txt', encoding='utf-8').read()
    assert txt.count('location:') == 129 and 'important features: w' in txt
