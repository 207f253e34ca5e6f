"""ISA statistics of the kernels in a hipcc -S listing:  python tools/isa_stats.py <file.s> [name-substring]
(build the listing with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isample_factory_amd/csrc -S
 --cuda-device-only sample_factory_amd/csrc/sf_nn.hip -o /tmp/nn.s)"""
import re
import sys


def kernel_stats(s: str, pat: str = ""):
    """[{name, vgpr, agpr, lds, scratch, mfma, glds, hot: {lines, mfma, glds, ds_read, waitcnt, barrier, valu, salu}}] for
    every kernel of the listing `s` whose mangled name contains `pat`"""
    out = []
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
        num = lambda v: int(v) if str(v).isdigit() else None
        out.append(dict(name=name, vgpr=num(get('num_vgpr')), agpr=num(get('num_agpr')), lds=num(lds.group(1)) if lds else None,
                        scratch=cnt('scratch_'), mfma=cnt('v_mfma'), glds=cnt('global_load_lds'), gload=cnt('global_load_dword'),
                        bload=cnt('buffer_load'),
                        hot=dict(lines=len(hot.splitlines()), mfma=h('v_mfma'), glds=h('global_load_lds'), gload=h('global_load_dword'),
                                 ds_read=h('ds_read'), ds_write=h('ds_write'), waitcnt=h('s_waitcnt'), barrier=h('s_barrier'),
                                 valu=h(r'\tv_(?!mfma)'), salu=h(r'\ts_(?!waitcnt|barrier|nop)'))))
    return out


def main():
    s = open(sys.argv[1]).read()
    for k in kernel_stats(s, sys.argv[2] if len(sys.argv) > 2 else ""):
        h = k["hot"]
        print(f"{k['name'][:70]}\n   vgpr {k['vgpr']} agpr {k['agpr']} sgpr ? lds {k['lds']} "
              f"scratch {k['scratch']} | total: mfma {k['mfma']} glds {k['glds']} gload {k['gload']} "
              f"bload {k['bload']} | hot block: {h['lines']} lines, mfma {h['mfma']} glds {h['glds']} "
              f"gload {h['gload']} ds_read {h['ds_read']} ds_write {h['ds_write']} waitcnt {h['waitcnt']} "
              f"barrier {h['barrier']} valu {h['valu']} salu {h['salu']}")


if __name__ == "__main__":
    main()
