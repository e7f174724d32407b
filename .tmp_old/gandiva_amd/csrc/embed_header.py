"""Embeds a header as a C++ string constant (gdv::gdv_device_lib_src) for hipRTC."""
import sys

src = open(sys.argv[1]).read()
assert ')GDVLIB"' not in src
# split into chunks: some compilers cap a single string literal's length
chunks = [src[i:i + 8000] for i in range(0, len(src), 8000)]
with open(sys.argv[2], "w") as f:
    f.write("// generated from %s by embed_header.py -- do not edit\n" % sys.argv[1])
    f.write("namespace gdv {\nextern const char gdv_device_lib_src[];\nconst char gdv_device_lib_src[] =\n")
    for c in chunks:
        f.write('R"GDVLIB(' + c + ')GDVLIB"\n')
    f.write(";\n}  // namespace gdv\n")
