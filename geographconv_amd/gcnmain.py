#!/usr/bin/env python
"""Train / evaluate entry point -- drop-in for /root/reference/gcnmain.py on MI355X.

Same flags (reference gcnmain.py:264-300), same flow (gcnmain.py:162-232): stack X, build the
index vectors, GraphConv.build_model / fit / predict, geolocation metrics.  Two things differ:
  * the data step.  The reference builds (A, X, Y, ...) from raw tweets with DataLoader
    (gcnmain.py:100-157; out of scope here, raw data withdrawn) or loads the 13-tuple from
    <dir>/dump.pkl (gcnmain.py:92-98).  This entry point loads a dump.pkl if there is one, and
    otherwise accepts `--synthetic cmu|twus` (the pinned generators of SURVEY.md §8d).
  * `--ngpu` / torchrun: one process per GPU, rows of the graph partitioned across ranks.
"""
from __future__ import annotations

import argparse
import gzip
import logging
import os
import pickle
import sys

import numpy as np
import scipy.sparse as sps

from .gcnmodel import GraphConv

logging.basicConfig(format='%(asctime)s %(message)s', datefmt='%m/%d/%Y %I:%M:%S %p', level=logging.INFO)
np.random.seed(77)          # reference gcnmain.py:39
model_args = None


# ---- data.py:28-34 ------------------------------------------------------------------------------
def dump_obj(obj, filename, protocol=-1, serializer=pickle):
    with gzip.open(filename, 'wb') as fout:
        serializer.dump(obj, fout, protocol)


class _OldDumpUnpickler(pickle.Unpickler):
    """The dumps the reference's users hold were written by Python 2.7 with the numpy / scipy of 2017 (README.md:32): their sparse
    matrices name `scipy.sparse.csr` / `.csc` / `.coo` ... -- module paths scipy keeps only as deprecated aliases (gone in SciPy 2.0).
    Every class of those modules lives in the public `scipy.sparse` namespace."""

    def find_class(self, module, name):
        if module.startswith('scipy.sparse.') and not module.startswith('scipy.sparse._') and hasattr(sps, name):
            return getattr(sps, name)
        return super().find_class(module, name)


def load_obj(filename, serializer=pickle):
    with gzip.open(filename, 'rb') as fin:
        if serializer is not pickle:
            return serializer.load(fin)
        try:
            return _OldDumpUnpickler(fin).load()
        except UnicodeDecodeError:               # python-2 pickles (README.md:32): 8-bit strings, raw array bytes among them
            fin.seek(0)
            return _OldDumpUnpickler(fin, encoding='latin1').load()


EARTH_RADIUS_KM = 6371.0088          # mean earth radius used by the `haversine` package the reference imports (gcnmain.py:20)


def haversine_km(lat1, lon1, lat2, lon2):
    """Great-circle distances in km, elementwise over arrays of degrees."""
    lat1, lon1, lat2, lon2 = (np.radians(np.asarray(v, dtype=np.float64)) for v in (lat1, lon1, lat2, lon2))
    h = np.sin((lat2 - lat1) * 0.5) ** 2 + np.cos(lat1) * np.cos(lat2) * np.sin((lon2 - lon1) * 0.5) ** 2
    return 2.0 * EARTH_RADIUS_KM * np.arcsin(np.sqrt(h))


def haversine(p1, p2):
    """Scalar form with the signature of the package function: haversine((lat, lon), (lat, lon)) -> km."""
    return float(haversine_km(p1[0], p1[1], p2[0], p2[1]))


def geo_eval(y_true, y_pred, U_eval, classLatMedian, classLonMedian, userLocation):
    """Geolocation metrics of a set of predictions (what reference gcnmain.py:43-63 reports): distance between each
    user's true coordinates ('lat,lon' strings in userLocation) and the median coordinates of the predicted class;
    -> (mean km, median km, Acc@161, distances, [[lat, lon]] true, [[lat, lon]] predicted) and the same log line."""
    n = len(y_pred)
    assert n == len(U_eval), "#preds: %d, #users: %d" % (n, len(U_eval))
    true = np.array([userLocation[u].split(',')[:2] for u in U_eval], dtype=np.float64).reshape(n, 2)
    labels = [str(c) for c in y_pred]
    pred = np.array([(classLatMedian[c], classLonMedian[c]) for c in labels], dtype=np.float64).reshape(n, 2)
    km = haversine_km(true[:, 0], true[:, 1], pred[:, 0], pred[:, 1])
    mean, median = np.mean(km), np.median(km)
    acc_at_161 = 100 * float(np.count_nonzero(km < 161)) / float(n)
    logging.info("Mean: " + str(int(mean)) + " Median: " + str(int(median)) + " Acc@161: " + str(int(acc_at_161)))
    return mean, median, acc_at_161, km.tolist(), true.tolist(), pred.tolist()


def synthetic_data(shape_name):
    """A dump.pkl-shaped 13-tuple (gcnmain.py:153) from the pinned generators: class medians and
    user locations are synthetic lat/lon so that geo_eval runs end to end."""
    from . import synth
    A, X, Y, (tr, dev, te), C = synth.make_graph(shape_name)
    rng = np.random.RandomState(5)
    cls_lat = rng.uniform(25, 49, C)
    cls_lon = rng.uniform(-124, -67, C)
    classLatMedian = {str(c): float(cls_lat[c]) for c in range(C)}
    classLonMedian = {str(c): float(cls_lon[c]) for c in range(C)}
    users = ['u%d' % i for i in range(X.shape[0])]
    lat = cls_lat[Y] + rng.normal(0, 0.5, len(Y))
    lon = cls_lon[Y] + rng.normal(0, 0.5, len(Y))
    userLocation = {u: '%f,%f' % (la, lo) for u, la, lo in zip(users, lat, lon)}
    sl = lambda idx: ([users[i] for i in idx])
    return (A, X[tr], Y[tr], X[dev], Y[dev], X[te], Y[te], sl(tr), sl(dev), sl(te), classLatMedian, classLonMedian,
            userLocation)


def preprocess_data(data_home, **kwargs):
    dump_file = os.path.join(data_home, 'dump.pkl')
    if os.path.exists(dump_file) and not model_args.builddata:
        logging.info('loading data from dumped file...')
        data = load_obj(dump_file)
        logging.info('loading data finished!')
        return data
    if getattr(model_args, 'synthetic', None):
        logging.info('generating the pinned synthetic %s-shape graph...' % model_args.synthetic)
        return synthetic_data(model_args.synthetic)
    raise FileNotFoundError(
        "%s not found. Building the dataset from raw tweets (reference DataLoader, gcnmain.py:100-157) "
        "is outside this package; provide the preprocessed dump.pkl (README of the reference) or run with "
        "--synthetic cmu|twus." % dump_file)


def setup_distributed(args):
    """-> (device, comm factory) for this process.  torchrun sets RANK / WORLD_SIZE / LOCAL_RANK."""
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return torch.device('cuda', 0), None
    from . import dist as gdist
    return gdist.init_process_group(int(os.environ.get('LOCAL_RANK', '0'))), 'dist'


class Splits:
    """The 13-tuple of dump.pkl (reference gcnmain.py:153,170) as the training loop wants it: one stacked float32 CSR X,
    one int32 label vector, float32 CSR A, and the row ranges of the train / dev / test users inside them."""

    def __init__(self, data, dtype='float32', dtypeint='int32'):
        (A, X_train, Y_train, X_dev, Y_dev, X_test, Y_test, self.U_train, self.U_dev, self.U_test,
         self.classLatMedian, self.classLonMedian, self.userLocation) = data
        self.dtypeint = dtypeint
        parts_x, parts_y = [X_train, X_dev, X_test], (Y_train, Y_dev, Y_test)
        self.X = sps.vstack(parts_x).tocsr().astype(dtype)
        stack = np.hstack if np.ndim(Y_train) == 1 else np.vstack
        self.Y = stack(parts_y).astype(dtypeint)
        self.A = sps.csr_matrix(A).astype(dtype)
        self.Y_dev, self.Y_test = Y_dev, Y_test
        n_tr, n_dev, n_te = (m.shape[0] for m in parts_x)
        self.n_train = n_tr
        self.all_train = np.arange(0, n_tr).astype(dtypeint)
        self.dev_indices = np.arange(n_tr, n_tr + n_dev).astype(dtypeint)
        self.test_indices = np.arange(n_tr + n_dev, n_tr + n_dev + n_te).astype(dtypeint)
        self.input_size = self.X.shape[1]
        self.output_size = int(np.max(self.Y) + 1)

    def draw_training_subset(self, fraction):
        """`-lblfraction`: a random subset of the training users, drawn from the global numpy stream exactly as the
        reference draws it (gcnmain.py:206-207: one np.random.choice without replacement per fraction)."""
        size = min(int(fraction * self.X.shape[0]), self.all_train.shape[0])
        return np.random.choice(self.all_train, size=size, replace=False).astype(self.dtypeint)

    def evaluate(self, clf, which):
        """predict + geo_eval on the dev or test users -> geo_eval's tuple."""
        idx, y, users = ((self.dev_indices, self.Y_dev, self.U_dev) if which == 'dev'
                         else (self.test_indices, self.Y_test, self.U_test))
        logging.info('%s results:' % which)
        y_pred, _ = clf.predict(self.X, self.A, idx)
        return geo_eval(y, y_pred, users, self.classLatMedian, self.classLonMedian, self.userLocation)


def run_fraction(clf, ds, fraction, args, batch_size, verbose, rank0):
    """One `-lblfraction` round: train (or load) on a fresh subset, report dev / test metrics, write the pickles the
    reference writes (model file if -save, dev distances always)."""
    logging.info('***********percentile %f ******************' % fraction)
    model_file = './data/model-{}-{}.pkl'.format(ds.A.shape[0], fraction)
    train_indices = ds.draw_training_subset(fraction)
    logging.info('{} training samples'.format(train_indices.shape[0]))
    if args.load:
        clf.load(load_obj, model_file)
        return train_indices, None
    if clf.fitted:
        clf.reset()             # parameters back to their initial values; Adam's state lives on, as in the reference
    clf.fit(ds.X, ds.A, ds.Y, train_indices=train_indices, val_indices=ds.dev_indices, n_epochs=args.epochs,
            batch_size=batch_size, max_down=args.maxdown, verbose=verbose and rank0, seed=model_args.seed)
    if args.save and rank0:
        os.makedirs(os.path.dirname(model_file), exist_ok=True)
        clf.save(dump_obj, model_file)
    dev = ds.evaluate(clf, 'dev')
    if rank0:
        with open('gcn_{}_percent_pred_{}.pkl'.format(fraction, ds.output_size), 'wb') as fout:
            pickle.dump(dev[3:], fout)                  # (distances, latlon_true, latlon_pred)
    test = ds.evaluate(clf, 'test')
    return train_indices, {'fraction': fraction, 'dev': dev[:3], 'test': test[:3]}


def main(data, args, **kwargs):
    logging.info('stacking training, dev and test features and creating indices...')
    ds = Splits(data)
    if args.vis:
        raise NotImplementedError("-vis (t-SNE plots, gcnmain.py:180-183) is outside the hot path")
    logging.info('running mlp with graph conv...')
    device, mode = setup_distributed(args)
    comm = None
    if mode == 'dist':
        from .dist import TorchDistComm
        comm = TorchDistComm(ds.X.shape[0], device)
    rank0 = comm is None or comm.rank == 0
    clf = GraphConv(input_size=ds.input_size, output_size=ds.output_size, hid_size_list=kwargs.get('hidden', [100]),
                    regul_coef=kwargs.get('regularization', 1e-6), drop_out=kwargs.get('dropout', 0.0),
                    batchnorm=args.batchnorm, highway=model_args.highway, device=device, comm=comm,
                    gemm_precision=getattr(args, 'gemm_precision', None), reorder=getattr(args, 'reorder', None))
    clf.build_model(ds.A, use_text=args.notxt, use_labels=args.lp, seed=model_args.seed)
    results, train_indices = [], None
    for fraction in args.lblfraction:
        train_indices, res = run_fraction(clf, ds, fraction, args, kwargs.get('batch', 500), not args.silent, rank0)
        if res is not None:
            results.append(res)
    if args.feature_report:
        report_features(clf, ds, train_indices, args, single_gpu=rank0 and comm is None)
    return clf, results


def report_features(clf, ds, train_indices, args, single_gpu):
    """`-feature_report` (reference gcnmain.py:234-246): probe the trained model with one one-hot "document" per
    vocabulary entry on an identity graph."""
    vocab_file = os.path.join(args.dir, 'vocab.pkl')
    if os.path.exists(vocab_file):
        vocab = load_obj(vocab_file)
    elif getattr(model_args, 'synthetic', None):
        vocab = {'w%d' % i: i for i in range(ds.X.shape[1])}
    else:
        logging.error('vocab file {} not found'.format(vocab_file))
        return
    logging.info('{} vocab loaded from file'.format(len(vocab)))
    if not single_gpu:
        logging.warning('-feature_report runs on a single GPU only')
        return
    # words seen >= 10 times among the training users are excluded from the report; dev vocabulary as the reference passes it
    seen = np.bincount(ds.X[train_indices].indices, minlength=ds.X.shape[1])
    train_vocab = set(np.nonzero(seen >= 10)[0].tolist())
    dev_vocab = set(np.unique(ds.X[ds.dev_indices].indices).tolist())
    eye = sps.identity(len(vocab), dtype='float32', format='csr')
    feature_report(clf, vocab, eye, eye, ds.classLatMedian, ds.classLonMedian, train_vocab, dev_vocab, topk=200,
                   dtypeint=ds.dtypeint)


def feature_report(model, vocab, X, A, classLatMedian, classLonMedian, train_vocab=set(), dev_vocab=set(), topk=20,
                   dtypeint='int32', filename='important_features.txt'):
    """Per class, the `topk` vocabulary entries whose one-hot document the model assigns to that class with the highest
    probability, skipping `train_vocab` (reference gcnmain.py:249-261; same output file format)."""
    import codecs
    _, probs = model.predict(X, A, np.arange(X.shape[0]).astype(dtypeint))
    word_of = {i: w for w, i in vocab.items()}
    logging.info('{} train vocab are being excluded!'.format(len(train_vocab)))
    ranking = np.argsort(-probs, axis=0)                       # column c: vocabulary ids by decreasing P(class c)
    allowed = np.ones(probs.shape[0], dtype=bool)
    allowed[[i for i in train_vocab if i < len(allowed)]] = False
    with codecs.open(filename, 'w', encoding='utf-8') as fout:
        for c in range(probs.shape[1]):
            order = ranking[:, c]
            words = ' '.join(word_of[int(i)] for i in order[allowed[order]][:topk])
            fout.write(u'location: {},{} \nimportant features: {} \n\n'.format(classLatMedian[str(c)], classLonMedian[str(c)],
                                                                              words))
    logging.info('important features are written to {}'.format(filename))
    return ranking


def parse_args(argv):
    """Reference flag set (gcnmain.py:271-298) + --synthetic / --epochs."""
    parser = argparse.ArgumentParser()
    parser.add_argument('-i', '--dataset', metavar='str', help='dataset for dialectology', type=str, default='na')
    parser.add_argument('-bucket', '--bucket', metavar='int', help='discretisation bucket size', type=int, default=300)
    parser.add_argument('-batch', '--batch', metavar='int', help='SGD batch size', type=int, default=500)
    parser.add_argument('-hid', nargs='+', type=int, help="list of hidden layer sizes", default=[100])
    parser.add_argument('-mindf', '--mindf', metavar='int', help='minimum document frequency in BoW', type=int, default=10)
    parser.add_argument('-d', '--dir', metavar='str', help='home directory', type=str, default='./data')
    parser.add_argument('-enc', '--encoding', metavar='str', help='Data Encoding (e.g. latin1, utf-8)', type=str, default='utf-8')
    parser.add_argument('-reg', '--regularization', metavar='float', help='regularization coefficient)', type=float, default=1e-6)
    parser.add_argument('-cel', '--celebrity', metavar='int', help='celebrity threshold', type=int, default=10)
    parser.add_argument('-conv', '--convolution', action='store_true', help='if true do convolution')
    parser.add_argument('-tune', '--tune', action='store_true', help='if true tune the hyper-parameters')
    parser.add_argument('-tf', '--tensorflow', action='store_true', help='if exists run with tensorflow')
    parser.add_argument('-batchnorm', action='store_true', help='if exists do batch normalization')
    parser.add_argument('-dropout', type=float, help="dropout value default(0)", default=0)
    parser.add_argument('-percent', action='store_true', help='if exists loop over different train/dev proportions')
    parser.add_argument('-vis', metavar='str', help='visualise representations', type=str, default=None)
    parser.add_argument('-builddata', action='store_true', help='if exists do not reload dumped data, build it from scratch')
    parser.add_argument('-lp', action='store_true', help='if exists use label information')
    parser.add_argument('-notxt', action='store_false', help='if exists do not use text information')
    parser.add_argument('-maxdown', help='max iter for early stopping', type=int, default=10)
    parser.add_argument('-silent', action='store_true', help='if exists be silent during training')
    parser.add_argument('-highway', action='store_true', help='if exists use highway connections else do not')
    parser.add_argument('-seed', metavar='int', help='random seed', type=int, default=77)
    parser.add_argument('-save', action='store_true', help='if exists save the model after training')
    parser.add_argument('-load', action='store_true', help='if exists load pretrained model from file')
    parser.add_argument('-feature_report', action='store_true', help='if exists report the important features of each location')
    parser.add_argument('-lblfraction', nargs='+', type=float, help="fraction of labelled data used for training e.g. 0.01 0.1", default=[1.0])
    # additions
    parser.add_argument('--synthetic', choices=['cmu', 'twus'], default=None, help='use the pinned synthetic graph instead of dump.pkl')
    parser.add_argument('--gemm-precision', choices=['f32', 'bf16x3', 'bf16'], default=None,
                        help="how H.W products are formed: f32 = exact fp32 MFMA (default), bf16x3 = 3-term bf16 split, bf16 = BASELINE config 5")
    parser.add_argument('--reorder', choices=['auto', 'lpa', 'rcm', 'bfs', 'degree'], default=None,
                        help="renumber the nodes on the device side (results stay in original node order); auto = label "
                             "propagation when it makes the graph product's gathers local, else nothing")
    parser.add_argument('--epochs', type=int, default=10000, help='max epochs (reference hard-codes 10000, gcnmain.py:221)')
    return parser.parse_args(argv)


def run(argv):
    global model_args
    args = parse_args(argv)
    model_args = args
    data = preprocess_data(data_home=args.dir, encoding=args.encoding, celebrity=args.celebrity, bucket=args.bucket,
                           mindf=args.mindf)
    return main(data, args, batch=args.batch, hidden=args.hid, regularization=args.regularization,
                dropout=args.dropout, percent=args.percent)


if __name__ == '__main__':
    run(sys.argv[1:])
