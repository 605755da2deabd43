"""Small host utilities (interface of mdir/tools/utils.py: indent :11-12, splitp :94-112, fs access of
mdir/external/daan/data/fs_driver.py:177-183 reduced to what the hub path needs)."""
import io
import pickle
from urllib import request


def indent(string, indent=1):
    return string.replace("\n", "\n" + "    " * indent)


def splitp(seq, sep, pairs=("()", "[]", "{}"), check_valid_pairs=False):
    """Split ``seq`` on ``sep`` but never inside a bracket pair: 'a:[1,2],b' -> ['a:[1,2]', 'b']."""
    opening = {p[0]: p[1] for p in pairs}
    stack, parts = [], [""]
    for ch in seq:
        if ch == sep and not stack:
            parts.append("")
            continue
        if ch in opening:
            stack.append(opening[ch])
        elif stack and ch == stack[-1]:
            stack.pop()
        parts[-1] += ch
    if check_valid_pairs:
        assert not stack, 'Invalid seq "%s": unbalanced %s' % (seq, stack)
    return parts


def fs_open(path):
    """Binary read handle for a local path or an http(s) URL (the reference goes through daan's fs_driver)."""
    if path.startswith("http://") or path.startswith("https://"):
        with request.urlopen(path) as resp:       # no network on the build / GPU boxes: raises URLError like the reference
            return io.BytesIO(resp.read())
    return open(path, "rb")


class _ArrayUnpickler(pickle.Unpickler):
    """The whitening files are ``{"P": ndarray, "m": ndarray}`` (mdir/stages/whiten.py:75): plain containers and numpy arrays only.
    Anything else in the stream (it may have come over plain http, hub/model.py BASE_URL) is refused instead of executed."""

    _ALLOWED = {("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy.core.numeric", "_frombuffer"),
                ("numpy._core.numeric", "_frombuffer"), ("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("refusing to unpickle %s.%s (only containers and numpy arrays are accepted)" % (module, name))


def fs_load_pickle(path):
    with fs_open(path) as handle:
        return _ArrayUnpickler(handle).load()
