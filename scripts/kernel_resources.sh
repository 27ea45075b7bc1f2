#!/bin/bash
# usage: scripts/kernel_resources.sh <file.hip> [grep pattern]   - VGPR / AGPR / SGPR / scratch / LDS of every gfx950 kernel in a source file
# (device-only compile into /tmp; the scratch column is the first thing to look at after touching an MFMA kernel: it must stay 0)
cd "$(dirname "$0")/../unpaired-image-generation_amd/csrc"
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc --offload-device-only -c $f -o /tmp/_kres.co || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/_kres.co --output=/tmp/_kres.elf --unbundle || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/_kres.elf | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    blk='.agpr_count'+blk
    g=lambda k: (re.search(k+r':\s+(\S+)', blk) or [None,'?'])[1]
    n=g(r'\.name')
    try: n=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',n],capture_output=True,text=True).stdout.strip()
    except Exception: pass
    print('vgpr %3s agpr %3s sgpr %3s scratch %4s lds %6s  %s' % (g(r'\.vgpr_count'),g(r'\.agpr_count'),g(r'\.sgpr_count'),g(r'\.private_segment_fixed_size'),g(r'\.group_segment_fixed_size'),n[:150]))
" | grep -E "$pat"
