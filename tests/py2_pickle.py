"""Test helper: write a pickle the way CPython 2.7 with the numpy / scipy of its time did (the reference is a Python-2 program and the
dump.pkl files its users hold are Python-2 pickles, reference README.md:32): protocol 2, 8-bit `str` objects (SHORT_BINSTRING / BINSTRING)
for ASCII text AND for raw array bytes, `unicode` only for non-ASCII text, the module paths numpy and scipy.sparse had then.  There is no
Python 2 in the image, so the stream is produced by the pure-Python pickler with those three differences put back."""
import io
import pickle
import struct
import types


class Py2StylePickler(pickle._Pickler):
    dispatch = dict(pickle._Pickler.dispatch)
    OLD = {'numpy._core.multiarray': 'numpy.core.multiarray', 'numpy._core.numeric': 'numpy.core.numeric',
           'scipy.sparse._csr': 'scipy.sparse.csr', 'scipy.sparse._csc': 'scipy.sparse.csc', 'scipy.sparse._coo': 'scipy.sparse.coo'}

    def _binstring(self, obj, b):
        n = len(b)
        self.write((pickle.SHORT_BINSTRING + bytes([n]) if n < 256 else pickle.BINSTRING + struct.pack('<i', n)) + b)
        self.memoize(obj)

    def save_str(self, obj):
        try:
            self._binstring(obj, obj.encode('ascii'))
        except UnicodeEncodeError:
            pickle._Pickler.save_str(self, obj)          # (a Python-2 `unicode`)
    dispatch[str] = save_str

    def save_bytes(self, obj):
        self._binstring(obj, obj)
    dispatch[bytes] = save_bytes

    def save_global(self, obj, name=None):
        mod = getattr(obj, '__module__', None)
        if mod in self.OLD:
            self.write(pickle.GLOBAL + self.OLD[mod].encode() + b'\n' + (name or obj.__qualname__).encode() + b'\n')
            self.memoize(obj)
            return
        pickle._Pickler.save_global(self, obj, name)
    dispatch[type] = save_global
    dispatch[types.FunctionType] = save_global
    dispatch[types.BuiltinFunctionType] = save_global


def dumps(obj):
    f = io.BytesIO()
    Py2StylePickler(f, protocol=2, fix_imports=True).dump(obj)
    return f.getvalue()
