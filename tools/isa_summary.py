"""tools/isa_summary.py <file.s> <kernel-name-substring>: registers, scratch and instruction counts of the matching kernels
in `hipcc -S --cuda-device-only` output (what DESIGN section 5 asks to look at after every kernel change)."""
import re
import sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\S*%s\S*):" % re.escape(sys.argv[2]), s, re.M):
    name = m.group(1)
    body = s[m.end():s.index("s_endpgm", m.end())]
    meta = s[s.index(".amdhsa_kernel " + name):]
    meta = meta[:meta.index(".end_amdhsa_kernel")]
    g = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, meta).group(1)
    cnt = lambda pat: len(re.findall(pat, body))
    print(name[:70], "vgpr", g("next_free_vgpr"), "sgpr", g("next_free_sgpr"), "scratch", g("private_segment_fixed_size"),
          "| readlane", cnt("v_readlane"), "ds_read", cnt("ds_read"), "ds_write", cnt("ds_write"), "lds-dma", cnt(r"global_load_lds"),
          "s_load", cnt("s_load"), "global_load", cnt(r"global_load_(?!lds)"), "flat", cnt(r"\bflat_"), "pk_fma", cnt("v_pk_fma"),
          "fma", cnt("v_fma_f32|v_fmac_f32"), "mfma", cnt("v_mfma"), "waitcnt", cnt("s_waitcnt"), "lines", body.count("\n"))
