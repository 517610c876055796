# generates the two specialisations of EncPx<P> (asm on the private registers of k_encode); HDUP=1: no look-up for a pixel whose table
# address equals its left neighbour's (same lane, same 4-pixel row segment)
import sys
HDUP = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def regs(base): return ["v%d" % (base + k) for k in range(16)]
def quads(base): return ["v[%d:%d]" % (base + 4 * i, base + 4 * i + 3) for i in range(4)]
out = []
for P, base in ((0, 96), (1, 112)):
    r = regs(base); q = quads(base)
    clob = ", ".join('"%s"' % x for x in r)
    L = []
    L.append("template <> struct EncPx<%d> {" % P)
    L.append("\tstatic __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, const uint32_t (&o)[4])")
    L.append("\t{")
    L.append('\t\tasm volatile("s_nop 4\\n\\t"\n\t\t    ' + "\n\t\t    ".join('"buffer_load_dwordx4 %s, %%%d, %%0, 0 offen nt%s"' % (q[i], 1 + i, "\\n\\t" if i < 3 else "") for i in range(4)))
    L.append('\t\t    :: "s"(rs), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]) : "memory", %s);' % clob)
    L.append("\t}")
    # table offsets straight from the set (lut_offset(), five instructions per pixel)
    L.append("\tstatic __device__ __forceinline__ void offsets(uint32_t (&t)[16])")
    L.append("\t{")
    L.append("\t\tuint32_t h;")
    lines = []
    for k in range(16):
        lines.append('"v_and_b32 %%%d, 0x030303, %s\\n\\tv_and_b32 %%16, 0xfcfcfc, %s\\n\\tv_mul_u32_u24 %%%d, 0x10410, %%%d\\n\\t"' % (k, r[k], r[k], k, k))
        lines.append('"v_bfe_u32 %%%d, %%%d, 15, 7\\n\\tv_lshl_or_b32 %%%d, %%16, 5, %%%d%s"' % (k, k, k, k, "\\n\\t" if k < 15 else ""))
    L.append("\t\tasm volatile(" + "\n\t\t    ".join(lines))
    L.append("\t\t    : " + ", ".join('"=&v"(t[%d])' % k for k in range(16)) + ', "=&v"(h) :: "memory");')
    L.append("\t}")
    L.append("\tstatic __device__ __forceinline__ void take_plain(uint32_t (&n)[16], uint32_t mask)")
    L.append("\t{")
    L.append("\t\tasm volatile(" + "\n\t\t    ".join('"v_and_b32 %%%d, %%16, %s%s"' % (k, r[k], "\\n\\t" if k < 15 else "") for k in range(16)))
    L.append("\t\t    : " + ", ".join('"=&v"(n[%d])' % k for k in range(16)) + ' : "v"(mask) : "memory");')
    L.append("\t}")
    L.append("\tstatic __device__ __forceinline__ void lookup_issue_wait(uint32_t (&e)[16], const uint32_t (&t)[16], __amdgpu_buffer_rsrc_t lut,")
    L.append("\t                                                         __amdgpu_buffer_rsrc_t rs, const uint32_t (&o)[4])")
    L.append("\t{")
    lines = ['"s_nop 4\\n\\t"']
    if HDUP:
        for k in (0, 4, 8, 12): lines.append('"buffer_load_ushort %%%d, %%%d, %%32, 0 offen\\n\\t"' % (k, 16 + k))
        for k in range(16):
            if k % 4 == 0: continue
            lines.append('"s_mov_b64 exec, -1\\n\\tv_cmp_ne_u32_e32 vcc, %%%d, %%%d\\n\\ts_mov_b64 exec, vcc\\n\\tbuffer_load_ushort %%%d, %%%d, %%32, 0 offen\\n\\t"' % (16 + k, 15 + k, k, 16 + k))
        lines.append('"s_mov_b64 exec, -1\\n\\t"')
    else:
        lines += ['"buffer_load_ushort %%%d, %%%d, %%32, 0 offen\\n\\t"' % (k, 16 + k) for k in range(16)]
    lines += ['"buffer_load_dwordx4 %s, %%%d, %%33, 0 offen nt\\n\\t"' % (q[i], 34 + i) for i in range(4)]
    if HDUP:
        lines.append('"s_waitcnt vmcnt(4)\\n\\t"')
        ks = [k for k in range(16) if k % 4]
        for j, k in enumerate(ks):
            lines.append('"v_cmp_eq_u32_e32 vcc, %%%d, %%%d\\n\\tv_cndmask_b32_e32 %%%d, %%%d, %%%d, vcc%s"' % (16 + k, 15 + k, k, k, k - 1, "\\n\\t" if j < len(ks) - 1 else ""))
    else:
        lines.append('"s_waitcnt vmcnt(4)"')
    L.append("\t\tasm volatile(" + "\n\t\t    ".join(lines))
    L.append("\t\t    : " + ", ".join('"=&v"(e[%d])' % k for k in range(16)))
    L.append("\t\t    : " + ", ".join('"v"(t[%d])' % k for k in range(16)) + ', "s"(lut), "s"(rs), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])')
    L.append('\t\t    : "memory", "vcc", %s);' % clob)
    L.append("\t}")
    L.append("\tstatic __device__ __forceinline__ void issue_wait(__amdgpu_buffer_rsrc_t rs, const uint32_t (&o)[4])")
    L.append("\t{")
    lines = ['"s_nop 4\\n\\t"'] + ['"buffer_load_dwordx4 %s, %%%d, %%0, 0 offen nt\\n\\t"' % (q[i], 1 + i) for i in range(4)]
    lines.append('"s_waitcnt vmcnt(4)"')
    L.append("\t\tasm volatile(" + "\n\t\t    ".join(lines))
    L.append('\t\t    :: "s"(rs), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]) : "memory", %s);' % clob)
    L.append("\t}")
    L.append("};")
    out.append("\n".join(L))
print("\n".join(out))
