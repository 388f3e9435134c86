#!/usr/bin/env python3
"""tools/gen_jpeg_golden.py -- writes tests/golden/jpeg_decode_sha256.json: the SHA-256 of libjpeg-turbo's (Pillow's) decode of the
reference's own JPEG files (tests/golden/*.jpg, copied unchanged from /root/reference/tests/).  The digests pin oracle/jpeg_ref.c's
LIBJPEG flavour independently of the Pillow build a later test run happens to find."""
import hashlib
import io
import json
import os

import numpy as np
import PIL
from PIL import Image, features

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
out = {}
for name in ["bench.jpg", "Prophecy_Has_Been_Fulfilled_1.jpg", "Prophecy_Has_Been_Fulfilled_2.jpg"]:
    data = open(os.path.join(root, name), "rb").read()
    a = np.array(Image.open(io.BytesIO(data)))
    out[name] = {"shape": list(a.shape), "libjpeg_turbo_rgb8_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                 "file_sha256": hashlib.sha256(data).hexdigest()}
out["_generator"] = {"pillow": PIL.__version__, "libjpeg_turbo": features.version("libjpeg_turbo")}
with open(os.path.join(root, "jpeg_decode_sha256.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
