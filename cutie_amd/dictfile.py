"""Reader/writer for the reference's ``dict/<prefix>_*.npy`` vocabulary files.

The reference stores python dicts with ``np.save(path, dict)`` and reads them back with
``np.load(path, allow_pickle=True).item()`` (data_loader_json.py:11-12, :79-81, :107-110).
Such a file is an ``.npy`` header for a 0-d object array followed by a pickle of the array.

Unpickling runs code named by the file.  This reader never does: it walks the pickle opcode
stream with ``pickletools.genops`` (a pure parser) and interprets only the data opcodes
(str/int/float/bytes/None/bool, tuple/list/dict construction, memo).  ``GLOBAL``/``REDUCE``/
``BUILD`` are kept as inert records so the object list of the numpy wrapper can be located;
nothing from the file is imported or called.  Anything outside that subset raises.

``*_classes.npy`` is a plain ``<U`` string array and loads with ``allow_pickle=False``.
"""
import io
import pickletools

import numpy as np
from numpy.lib import format as _npformat

__all__ = ["load_npy_dict", "save_npy_dict", "load_npy_classes", "load_dictionary_triplet"]


class _Inert(object):
    """Stand-in for something the pickle asked to import/call/build. Never executed."""
    __slots__ = ("kind", "payload", "state")

    def __init__(self, kind, payload):
        self.kind = kind
        self.payload = payload
        self.state = None

    def __repr__(self):
        return "<inert %s %r>" % (self.kind, self.payload if self.kind == "global" else "...")


_MARK = object()


def _pop_mark(stack):
    items = []
    while True:
        if not stack:
            raise ValueError("dict file: unbalanced MARK")
        top = stack.pop()
        if top is _MARK:
            break
        items.append(top)
    items.reverse()
    return items


def _interpret(data):
    """Tiny data-only pickle machine. Returns the top-of-stack object at STOP."""
    stack, memo = [], {}
    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "STOP":
            break
        elif n in ("BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8", "UNICODE",
                   "BININT", "BININT1", "BININT2", "INT", "LONG", "LONG1", "LONG4",
                   "BINFLOAT", "FLOAT", "BINBYTES", "SHORT_BINBYTES", "BINBYTES8",
                   "BINSTRING", "SHORT_BINSTRING", "STRING"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "MARK":
            stack.append(_MARK)
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "TUPLE1":
            stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2":
            stack[-2:] = [(stack[-2], stack[-1])]
        elif n == "TUPLE3":
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n == "TUPLE":
            stack.append(tuple(_pop_mark(stack)))
        elif n == "LIST":
            stack.append(list(_pop_mark(stack)))
        elif n == "DICT":
            it = _pop_mark(stack)
            stack.append(dict(zip(it[0::2], it[1::2])))
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "APPENDS":
            it = _pop_mark(stack)
            stack[-1].extend(it)
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            it = _pop_mark(stack)
            d = stack[-1]
            if not isinstance(d, dict):
                raise ValueError("dict file: SETITEMS on non-dict")
            for k, v in zip(it[0::2], it[1::2]):
                d[k] = v
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[arg] = stack[-1]
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[arg])
        elif n == "GLOBAL":
            stack.append(_Inert("global", arg))
        elif n == "STACK_GLOBAL":
            name = stack.pop()
            mod = stack.pop()
            stack.append(_Inert("global", "%s %s" % (mod, name)))
        elif n == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            stack.append(_Inert("reduce", (fn, args)))
        elif n == "BUILD":
            state = stack.pop()
            tgt = stack[-1]
            if not isinstance(tgt, _Inert):
                raise ValueError("dict file: BUILD on plain data")
            tgt.state = state
        else:
            raise ValueError("dict file: pickle opcode %s is outside the data-only subset" % n)
    if not stack:
        raise ValueError("dict file: empty pickle")
    return stack[-1]


def _read_object_payload(path):
    with open(path, "rb") as f:
        version = _npformat.read_magic(f)
        if version == (1, 0):
            shape, _fortran, dtype = _npformat.read_array_header_1_0(f)
        elif version == (2, 0):
            shape, _fortran, dtype = _npformat.read_array_header_2_0(f)
        else:
            raise ValueError("%s: unsupported npy version %r" % (path, version))
        if dtype != np.dtype("O"):
            raise ValueError("%s: not an object-array npy (dtype %s)" % (path, dtype))
        if shape != ():
            raise ValueError("%s: expected a 0-d object array, got shape %r" % (path, shape))
        return f.read()


def load_npy_dict(path):
    """Return the python dict saved by ``np.save(path, some_dict)`` without unpickling."""
    top = _interpret(_read_object_payload(path))
    # top is the inert ndarray record; its BUILD state is
    # (version, shape, dtype-record, is_fortran, [objects...])
    if not isinstance(top, _Inert) or not isinstance(top.state, tuple) or len(top.state) < 5:
        raise ValueError("%s: not a pickled 0-d numpy object array" % path)
    objs = top.state[4]
    if not isinstance(objs, list) or len(objs) != 1 or not isinstance(objs[0], dict):
        raise ValueError("%s: payload is not a single dict" % path)
    d = objs[0]
    for k, v in d.items():
        if not isinstance(k, (str, int)) or not isinstance(v, (str, int)):
            raise ValueError("%s: dict holds non str/int entries" % path)
    return d


def save_npy_dict(path, d):
    """Write ``d`` in the reference's format (data_loader_json.py:107-109)."""
    if not path.endswith(".npy"):
        path = path + ".npy"
    arr = np.empty((), dtype=object)
    arr[()] = dict(d)
    buf = io.BytesIO()
    np.save(buf, arr, allow_pickle=True)
    with open(path, "wb") as f:
        f.write(buf.getvalue())


def load_npy_classes(path):
    """``*_classes.npy`` is a unicode array; never needs pickle (data_loader_json.py:110)."""
    return [str(x) for x in np.load(path, allow_pickle=False)]


def load_dictionary_triplet(prefix):
    """(dictionary, word_to_index, index_to_word) as data_loader_json.py:79-81 loads them."""
    return (load_npy_dict(prefix + "_dictionary.npy"),
            load_npy_dict(prefix + "_word_to_index.npy"),
            load_npy_dict(prefix + "_index_to_word.npy"))
