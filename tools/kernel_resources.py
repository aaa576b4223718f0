import subprocess, sys, re
LL='/opt/rocm/lib/llvm/bin/'
def res(obj):
    subprocess.run(['objcopy','-O','binary','--only-section=.hip_fatbin',obj,'/tmp/_fat.bin'],check=True)
    out = subprocess.run([LL+'clang-offload-bundler','--list','--type=o','--input=/tmp/_fat.bin'],capture_output=True,text=True).stdout.split()
    tgt=[t for t in out if 'gfx950' in t][0]
    subprocess.run([LL+'clang-offload-bundler','--unbundle','--type=o','--input=/tmp/_fat.bin','--targets='+tgt,'--output=/tmp/_co.o'],check=True)
    txt = subprocess.run([LL+'llvm-readelf','--notes','/tmp/_co.o'],capture_output=True,text=True).stdout
    ks = {}; cur = {}
    for line in txt.splitlines():
        line=line.strip()
        if line.startswith('- .agpr_count') or line.startswith('- .args'):
            if cur.get('name'): ks[cur['name']] = cur
            cur = {}
        for key in ('.name:', '.vgpr_count:', '.sgpr_count:', '.agpr_count:', '.vgpr_spill_count:', '.group_segment_fixed_size:', '.private_segment_fixed_size:'):
            if key in line:
                cur[key.strip('.:')] = line.split(key)[1].strip()
    if cur.get('name'): ks[cur['name']] = cur
    return ks
a = res(sys.argv[1])
for n,k in sorted(a.items()):
    dn = subprocess.run(['c++filt', n],capture_output=True,text=True).stdout.strip()
    if len(sys.argv) > 2 and sys.argv[2] not in dn: continue
    print(dn[:120], '| vgpr', k.get('vgpr_count'), 'agpr', k.get('agpr_count'), 'spill', k.get('vgpr_spill_count'), 'scratch', k.get('private_segment_fixed_size'), 'sgpr', k.get('sgpr_count'))
