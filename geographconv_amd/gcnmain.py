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


def load_obj(filename, serializer=pickle):
    with gzip.open(filename, 'rb') as fin:
        try:
            return serializer.load(fin)
        except UnicodeDecodeError:               # python-2 pickles (README.md:32)
            fin.seek(0)
            return serializer.load(fin, encoding='latin1')


def haversine(p1, p2):
    """Great-circle distance in km (the `haversine` package the reference imports, gcnmain.py:20;
    mean earth radius 6371.0088 km)."""
    lat1, lon1 = np.radians(p1[0]), np.radians(p1[1])
    lat2, lon2 = np.radians(p2[0]), np.radians(p2[1])
    d = np.sin((lat2 - lat1) * 0.5) ** 2 + np.cos(lat1) * np.cos(lat2) * np.sin((lon2 - lon1) * 0.5) ** 2
    return float(2 * 6371.0088 * np.arcsin(np.sqrt(d)))


def geo_eval(y_true, y_pred, U_eval, classLatMedian, classLonMedian, userLocation):
    """Mean / median error km and Acc@161 (reference gcnmain.py:43-63)."""
    assert len(y_pred) == len(U_eval), "#preds: %d, #users: %d" % (len(y_pred), len(U_eval))
    distances, latlon_pred, latlon_true = [], [], []
    for i in range(0, len(y_pred)):
        user = U_eval[i]
        location = userLocation[user].split(',')
        lat, lon = float(location[0]), float(location[1])
        latlon_true.append([lat, lon])
        prediction = str(y_pred[i])
        lat_pred, lon_pred = classLatMedian[prediction], classLonMedian[prediction]
        latlon_pred.append([lat_pred, lon_pred])
        distances.append(haversine((lat, lon), (lat_pred, lon_pred)))
    acc_at_161 = 100 * len([d for d in distances if d < 161]) / float(len(distances))
    logging.info("Mean: " + str(int(np.mean(distances))) + " Median: " + str(int(np.median(distances))) +
                 " Acc@161: " + str(int(acc_at_161)))
    return np.mean(distances), np.median(distances), acc_at_161, distances, latlon_true, latlon_pred


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
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    return torch.device('cuda', local), 'dist'


def main(data, args, **kwargs):
    batch_size = kwargs.get('batch', 500)
    hidden_size = kwargs.get('hidden', [100])
    dropout = kwargs.get('dropout', 0.0)
    regul = kwargs.get('regularization', 1e-6)
    dtype = 'float32'
    dtypeint = 'int32'
    A, X_train, Y_train, X_dev, Y_dev, X_test, Y_test, U_train, U_dev, U_test, classLatMedian, classLonMedian, userLocation = data
    logging.info('stacking training, dev and test features and creating indices...')
    X = sps.vstack([X_train, X_dev, X_test]).tocsr()
    if len(Y_train.shape) == 1:
        Y = np.hstack((Y_train, Y_dev, Y_test))
    else:
        Y = np.vstack((Y_train, Y_dev, Y_test))
    Y = Y.astype(dtypeint)
    X = X.astype(dtype)
    A = sps.csr_matrix(A).astype(dtype)
    if args.vis:
        raise NotImplementedError("-vis (t-SNE plots, gcnmain.py:180-183) is outside the hot path")
    input_size = X.shape[1]
    output_size = int(np.max(Y) + 1)
    verbose = not args.silent
    fractions = args.lblfraction
    all_train_indices = np.asarray(range(0, X_train.shape[0])).astype(dtypeint)
    logging.info('running mlp with graph conv...')
    device, mode = setup_distributed(args)
    comm = None
    if mode == 'dist':
        from .dist import TorchDistComm
        comm = TorchDistComm(X.shape[0], device)
    rank0 = comm is None or comm.rank == 0
    clf = GraphConv(input_size=input_size, output_size=output_size, hid_size_list=hidden_size, regul_coef=regul,
                    drop_out=dropout, batchnorm=args.batchnorm, highway=model_args.highway, device=device, comm=comm,
                    gemm_precision=getattr(args, 'gemm_precision', None))
    clf.build_model(A, use_text=args.notxt, use_labels=args.lp, seed=model_args.seed)

    results = []
    for percentile in fractions:
        logging.info('***********percentile %f ******************' % percentile)
        model_file = './data/model-{}-{}.pkl'.format(A.shape[0], percentile)
        selection_size = min(int(percentile * X.shape[0]), all_train_indices.shape[0])
        train_indices = np.random.choice(all_train_indices, size=selection_size, replace=False).astype(dtypeint)
        logging.info('{} training samples'.format(train_indices.shape[0]))
        dev_indices = np.asarray(range(X_train.shape[0], X_train.shape[0] + X_dev.shape[0])).astype(dtypeint)
        test_indices = np.asarray(range(X_train.shape[0] + X_dev.shape[0],
                                        X_train.shape[0] + X_dev.shape[0] + X_test.shape[0])).astype(dtypeint)
        if args.load:
            clf.load(load_obj, model_file)
        else:
            if clf.fitted:
                clf.reset()
            clf.fit(X, A, Y, train_indices=train_indices, val_indices=dev_indices, n_epochs=args.epochs,
                    batch_size=batch_size, max_down=args.maxdown, verbose=verbose and rank0, seed=model_args.seed)
            if args.save and rank0:
                os.makedirs(os.path.dirname(model_file), exist_ok=True)
                clf.save(dump_obj, model_file)
            logging.info('dev results:')
            y_pred, _ = clf.predict(X, A, dev_indices)
            mean, median, acc, distances, latlon_true, latlon_pred = geo_eval(Y_dev, y_pred, U_dev, classLatMedian,
                                                                              classLonMedian, userLocation)
            if rank0:
                with open('gcn_{}_percent_pred_{}.pkl'.format(percentile, output_size), 'wb') as fout:
                    pickle.dump((distances, latlon_true, latlon_pred), fout)
            logging.info('test results:')
            y_pred, _ = clf.predict(X, A, test_indices)
            t = geo_eval(Y_test, y_pred, U_test, classLatMedian, classLonMedian, userLocation)
            results.append({'fraction': percentile, 'dev': (mean, median, acc), 'test': t[:3]})
    if args.feature_report:
        # gcnmain.py:234-246: probe the trained model with one-hot "documents" (X = I over the vocabulary) on an
        # identity graph and list, per class, the words it is most confident about
        vocab_file = os.path.join(args.dir, 'vocab.pkl')
        if os.path.exists(vocab_file):
            vocab = load_obj(vocab_file)
        elif getattr(model_args, 'synthetic', None):
            vocab = {'w%d' % i: i for i in range(X.shape[1])}
        else:
            logging.error('vocab file {} not found'.format(vocab_file))
            return clf, results
        logging.info('{} vocab loaded from file'.format(len(vocab)))
        from collections import Counter
        train_vocab = set(term for term, count in Counter(X[train_indices].nonzero()[1]).items() if count >= 10)
        dev_vocab = set(np.nonzero(np.asarray(X[dev_indices].sum(axis=0)).ravel())[0])
        X_onehot = sps.identity(len(vocab), dtype=dtype, format='csr')
        A_onehot = X_onehot
        if rank0 and comm is None:
            feature_report(clf, vocab, X_onehot, A_onehot, classLatMedian, classLonMedian, train_vocab, dev_vocab,
                           topk=200, dtypeint=dtypeint)
        else:
            logging.warning('-feature_report runs on a single GPU only')
    return clf, results


def feature_report(model, vocab, X, A, classLatMedian, classLonMedian, train_vocab=set(), dev_vocab=set(), topk=20,
                   dtypeint='int32', filename='important_features.txt'):
    """Top-k most indicative vocabulary entries per class (reference gcnmain.py:249-261, python-3 idioms)."""
    import codecs
    eval_indices = np.asarray(range(X.shape[0])).astype(dtypeint)
    preds, probs = model.predict(X, A, eval_indices)
    id2v = {v: k for k, v in vocab.items()}
    logging.info('{} train vocab are being excluded!'.format(len(train_vocab)))
    feature_importance = np.argsort(-probs, axis=0)
    with codecs.open(filename, 'w', encoding='utf-8') as fout:
        for lbl in range(probs.shape[1]):
            important_vocab = ' '.join([id2v[idx] for idx in feature_importance[:, lbl].reshape(-1).tolist()
                                        if idx not in train_vocab][0:topk])
            lat, lon = classLatMedian[str(lbl)], classLonMedian[str(lbl)]
            fout.write(u'location: {},{} \nimportant features: {} \n\n'.format(lat, lon, important_vocab))
    logging.info('important features are written to {}'.format(filename))
    return feature_importance


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
