"""Import stand-in for commentjson (absent from the image): JSON with // and # line comments."""
import json
import re


def load(handle, **kw):
    text = handle.read()
    text = re.sub(r'^\s*(//|#).*$', '', text, flags=re.M)
    return json.loads(text, **kw)
