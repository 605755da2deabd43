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


def fs_load_pickle(path):
    with fs_open(path) as handle:
        return pickle.load(handle)
