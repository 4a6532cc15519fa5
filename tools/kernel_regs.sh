#!/bin/bash
# usage: tools/kernel_regs.sh [libmi355pt.so]  -> VGPR / SGPR / scratch / LDS of every kernel in the library's gfx950 code objects
LIB=$(readlink -f ${1:-pbrt-rust_amd/csrc/libmi355pt.so})
T=$(mktemp -d)
cd $T
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$LIB >/dev/null 2>&1
python3 - "$LIB" <<'PY'
import re, subprocess, sys
data = open(sys.argv[1], 'rb').read()
# every embedded code object is an ELF for amdgcn: find them by the ELF magic + machine 0xE0 (EM_AMDGPU)
out = []
pos = 0
n = 0
while True:
    i = data.find(b'\x7fELF', pos)
    if i < 0: break
    pos = i + 4
    if data[i + 18:i + 20] != b'\xe0\x00': continue
    # e_shoff + e_shnum * e_shentsize gives the size
    import struct
    shoff = struct.unpack_from('<Q', data, i + 0x28)[0]; shentsize, shnum = struct.unpack_from('<HH', data, i + 0x3A)
    size = shoff + shentsize * shnum
    open(f'co{n}.elf', 'wb').write(data[i:i + size]); n += 1
for k in range(n):
    txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f'co{k}.elf'], capture_output=True, text=True).stdout
    for m in re.finditer(r'\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
        pass
    # the metadata is YAML: parse kernel blocks
    for blk in txt.split('- .agpr_count:')[1:]:
        g = lambda key: (re.search(r'\.%s:\s+(\S+)' % key, blk) or [None, '?'])[1]
        name = g('name')
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(.*', '', dem)
        out.append((dem, blk.split()[0], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
for o in sorted(set(out)):
    print('%-48s agpr %3s vgpr %3s sgpr %3s scratch %5s lds %6s' % o)
PY
rm -rf $T
