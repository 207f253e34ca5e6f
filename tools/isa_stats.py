"""ISA statistics of the kernels in a hipcc -S listing:  python tools/isa_stats.py <file.s> [name-substring]
(build the listing with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isample_factory_amd/csrc -S
 --cuda-device-only sample_factory_amd/csrc/sf_nn.hip -o /tmp/nn.s)"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pat not in name:
            continue
        ins = body.split('\n')
        cnt = lambda p: sum(1 for l in ins if re.search(p, l))
        get = lambda k: (re.search(r'\.set %s\.%s, (\d+)' % (re.escape(name), k), s) or [None, '?'])[1]
        lds = re.search(r'\.amdhsa_kernel %s\n.*?group_segment_fixed_size (\d+)' % re.escape(name), s, re.S)
        # main loop = the basic block with most MFMAs
        blocks = re.split(r'^\.LBB\d+_\d+:', body, flags=re.M)
        hot = max(blocks, key=lambda b: len(re.findall('v_mfma', b)))
        h = lambda p: len(re.findall(p, hot))
        valu, salu = h(r'\tv_(?!mfma)'), h(r'\ts_(?!waitcnt|barrier|nop)')
        print(f"{name[:70]}\n   vgpr {get('num_vgpr')} agpr {get('num_agpr')} sgpr {get('num_sgpr')} lds {lds and lds.group(1)} "
              f"scratch {cnt('scratch_')} | total: mfma {cnt('v_mfma')} glds {cnt('global_load_lds')} gload {cnt('global_load_dword')} "
              f"bload {cnt('buffer_load')} | hot block: {len(hot.splitlines())} lines, mfma {h('v_mfma')} glds {h('global_load_lds')} "
              f"gload {h('global_load_dword')} ds_read {h('ds_read')} ds_write {h('ds_write')} waitcnt {h('s_waitcnt')} "
              f"barrier {h('s_barrier')} valu {valu} salu {salu}")


if __name__ == "__main__":
    main()
