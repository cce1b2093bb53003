"""Writes tools/probe_mfma4c.hip: cycles per v_mfma_f64_4x4x4_4b_f64 in the instruction patterns of the small-basis
kernel (three accumulator chains per sub-step, 14 sub-steps per step), with the registers chosen by hand:
where do the ~20 cycles per MFMA of the kernel's steps come from when the instruction alone takes 16?
    python tools/gen_probe_mfma4c.py && hipcc -O3 --offload-arch=gfx950 tools/probe_mfma4c.hip -o /tmp/probe4c && /tmp/probe4c
"""
VARIANTS = {
    # name: (B operand file, accumulator file, extra instructions per sub-step, nop after the first MFMA, same A for the 3)
    "vgprB_agprC": ("v", "a", [], False, True),
    "agprB_agprC": ("a", "a", [], False, True),
    "agprB_agprC_nop": ("a", "a", [], True, True),
    "vgprB_vgprC": ("v", "v", [], False, True),
    "agprB_agprC_dsread": ("a", "a", ["dsr"], False, True),
    "vgprB_agprC_dsread": ("v", "a", ["dsr"], False, True),
    "vgprB_agprC_dsw_dsr": ("v", "a", ["dsw", "dsr"], False, True),
    "vgprB_agprC_load": ("v", "a", ["ld"], False, True),
    "vgprB_agprC_all": ("v", "a", ["dsw", "ld", "dsr"], False, True),
    "vgprB_agprC_diffA": ("v", "a", [], False, False),
    "agprB_agprC_diffA": ("a", "a", [], False, False),
    "vgprB_agprC_dsw": ("v", "a", ["dsw"], False, True),
    "vgprB_agprC_dsw32": ("v", "a", ["dsw32"], False, True),
    "vgprB_agprC_dsw_agpr": ("v", "a", ["dswa"], False, True),
    "vgprB_agprC_bufld": ("v", "a", ["bufld"], False, True),
    "vgprB_agprC_bufld4": ("v", "a", ["bufld4"], False, True),
    "vgprB_agprC_gld_saddr": ("v", "a", ["gsaddr"], False, True),
    "vgprB_agprC_bufst": ("v", "a", ["bufst"], False, True),
    "vgprB_agprC_dsr128": ("v", "a", ["dsr128"], False, True),
    "vgprB_agprC_bufld_dsw_dsr": ("v", "a", ["dsw", "bufld", "dsr"], False, True),
    "vgprB_agprC_bufldlds": ("v", "a", ["bufldlds"], False, True),
    "spread_dsw_dsr": ("v", "a", [["dsw"], ["dsr"], []], False, True),
    "spread_dsw_bufld_dsr": ("v", "a", [["dsw"], ["bufld"], ["dsr"]], False, True),
    "spread_ldslds_dsr": ("v", "a", [["bufldlds"], ["dsr"], []], False, True),
    "spread_ldslds4_dsr": ("v", "a", [["bufldlds4"], ["dsr"], []], False, True),
    "spread_ldslds_dsr_dsr": ("v", "a", [["bufldlds"], ["dsr"], ["dsr"]], False, True),
    "two_dsr_same_gap": ("v", "a", [["dsr", "dsr"], [], []], False, True),
    "two_dsr_two_gaps": ("v", "a", [["dsr"], ["dsr"], []], False, True),
    "three_dsr_three_gaps": ("v", "a", [["dsr"], ["dsr"], ["dsr"]], False, True),
    "bufld_every_other": ("v", "a", [["bufld_eo"], [], []], False, True),
    "nop7_each_gap": ("v", "a", [["nop7"], ["nop7"], ["nop7"]], False, True),
    "nop3_each_gap": ("v", "a", [["nop3"], ["nop3"], ["nop3"]], False, True),
    "valu_each_gap": ("v", "a", [["valu"], ["valu"], ["valu"]], False, True),
    "salu4_each_gap": ("v", "a", [["salu4"], ["salu4"], ["salu4"]], False, True),
    "bufst_own_addr": ("v", "a", [["bufst_own"], [], []], False, True),
    "bufst_own_agpr": ("v", "a", [["bufst_own_a"], [], []], False, True),
    "bufld2_every_other": ("v", "a", [["bufld_eo2"], [], []], False, True),
    "bufst4_own": ("v", "a", [["bufst4_own"], [], []], False, True),
    "bufst4_every_other": ("v", "a", [["bufst4_eo"], [], []], False, True),
    "bufst2_every_other": ("v", "a", [["bufst2_eo"], [], []], False, True),
    "bufst2_every_4th": ("v", "a", [["bufst2_e4"], [], []], False, True),
    "waitcnt_each_gap": ("v", "a", [["wait"], ["wait"], ["wait"]], False, True),
    "salu1_each_gap": ("v", "a", [["salu1"], ["salu1"], ["salu1"]], False, True),
    "phase1_like_x4": ("v", "a", [["bufld_eo", "dsw"], ["dsw"], ["dsr"]], False, True),
    "phase1_like_x4_waits": ("v", "a", [["bufld_eo", "wait", "dsw"], ["dsw"], ["wait", "dsr"]], False, True),
    "plan1": ("v", "a", [["bufld4@e", "dsr@o"], ["dsw@e", "dsr@o"], ["dsw@e"]], False, True),
    "plan1_b": ("v", "a", [["bufld4@e", "dsr@o"], ["dsw@e"], ["dsw@e", "dsr@o"]], False, True),
    "plan1_w2": ("v", "a", [["bufld4@e", "dsr@o"], ["dsw2@e", "dsr@o"], []], False, True),
    "plan1_lds4": ("v", "a", [["bufldlds4@e", "dsr@o"], ["dsr@o"], []], False, True),
    "bufld_dsr_same_gap": ("v", "a", [["bufld4@e", "dsr@e"], [], []], False, True),
    "dsw_dsw_same_gap": ("v", "a", [["dsw", "dsw"], [], []], False, True),
    "dsw_each_gap": ("v", "a", [["dsw"], ["dsw"], ["dsw"]], False, True),
    "phase2_plan": ("a", "a", [["dsr"], ["bufst2@e4"], []], False, True),
    "plan2": ("v", "a", [["dsw@e", "dsr@o"], ["dsw@e"], ["bufld4@e", "dsr@o"]], False, True),
    "plan3": ("v", "a", [["dsw@e", "dsr@o"], ["dsw@e", "dsr@o"], ["bufld4@e"]], False, True),
    "plan4": ("v", "a", [["dsw@e", "dsr@o"], ["dsw@e", "dsr@o"], ["bufld4@e", "dsr@o"]], False, True),
    "plan1_b_lf": ("v", "a", [["bufld4@e", "dsr@o"], ["dsw@e", "dsr@o"], ["dsw@e", "dsr@o"]], False, True),
    "plan5": ("v", "a", [["dsw@e"], ["dsw@e", "dsr@o"], ["bufld4@e", "dsr@o"]], False, True),
    "pair_vgprB": ("v", "a", [], False, "pair"),
    "pair_agprB": ("a", "a", [], False, "pair"),
}

def body(bfile, cfile, extras, nop, same_a):
    lines = []
    for ks in range(14):
        if same_a == "pair":      # two chains, A differs, B the same: the blocks of the last column group
            b = f"{bfile}[{40 + 2 * ks}:{41 + 2 * ks}]"
            lines.append(f"v_mfma_f64_4x4x4_4b_f64 {cfile}[200:201], v[{10 + 2 * ks}:{11 + 2 * ks}], {b}, {cfile}[200:201]")
            lines.append(f"v_mfma_f64_4x4x4_4b_f64 {cfile}[202:203], v[{150 + 2 * ks}:{151 + 2 * ks}], {b}, {cfile}[202:203]")
            continue
        for j in range(3):
            a = f"v[{10 + 2 * ks}:{11 + 2 * ks}]" if same_a else f"v[{10 + 2 * ((ks + 5 * j) % 14)}:{11 + 2 * ((ks + 5 * j) % 14)}]"
            b = f"{bfile}[{40 + 6 * ks + 2 * j}:{41 + 6 * ks + 2 * j}]"
            c = f"{cfile}[{200 + 2 * j}:{201 + 2 * j}]"
            lines.append(f"v_mfma_f64_4x4x4_4b_f64 {c}, {a}, {b}, {c}")
            if nop and j == 0:
                lines.append("s_nop 0")
            gap = extras[j] if (extras and isinstance(extras[0], list)) else (extras if j == 0 else [])
            if True:
                for e in gap:
                    if "@" in e:
                        e, when = e.split("@")
                        if when == "e" and ks % 2: continue
                        if when == "o" and not ks % 2: continue
                        if when == "e4" and ks % 4: continue
                    if e == "dsr":
                        lines.append(f"ds_read_b64 v[{130 + 2 * (ks % 4)}:{131 + 2 * (ks % 4)}], %2 offset:{512 * ks}")
                    if e == "dsw":
                        lines.append(f"ds_write_b64 %2, v[{140 + 2 * (ks % 4)}:{141 + 2 * (ks % 4)}] offset:{8192 + 512 * ks}")
                    if e == "dsw32":
                        lines.append(f"ds_write_b32 %2, v{140 + 2 * (ks % 4)} offset:{8192 + 512 * ks}")
                    if e == "dswa":
                        lines.append(f"ds_write_b64 %2, a[{100 + 2 * (ks % 4)}:{101 + 2 * (ks % 4)}] offset:{8192 + 512 * ks}")
                    if e == "bufld":
                        lines.append(f"buffer_load_dwordx2 v[{140 + 2 * (ks % 4)}:{141 + 2 * (ks % 4)}], %2, s[28:31], 0 offen offset:{256 * ks}")
                    if e == "bufld4":
                        lines.append(f"buffer_load_dwordx4 v[{140 + 4 * (ks % 2)}:{143 + 4 * (ks % 2)}], %2, s[28:31], 0 offen offset:{256 * ks}")
                    if e == "gsaddr":
                        lines.append(f"global_load_dwordx2 v[{140 + 2 * (ks % 4)}:{141 + 2 * (ks % 4)}], %2, s[28:29] offset:{256 * ks}")
                    if e == "bufst":
                        lines.append(f"buffer_store_dwordx2 v[{144 + 2 * (ks % 2)}:{145 + 2 * (ks % 2)}], %2, s[28:31], 0 offen offset:{256 * ks}")
                    if e == "dsr128":
                        lines.append(f"ds_read_b128 v[{130 + 4 * (ks % 2)}:{133 + 4 * (ks % 2)}], %2 offset:{512 * ks}")
                    if e == "bufldlds":
                        lines.append(f"s_mov_b32 m0, {8192 + 256 * ks}")
                        lines.append(f"buffer_load_dword %2, s[28:31], 0 offen offset:{256 * ks} lds")
                    if e == "bufldlds4":
                        lines.append(f"s_mov_b32 m0, {8192 + 1024 * (ks % 8)}")
                        lines.append(f"buffer_load_dwordx4 %2, s[28:31], 0 offen offset:{256 * ks} lds")
                    if e == "bufld_eo" and ks % 2 == 0:
                        lines.append(f"buffer_load_dwordx4 v[{140 + 4 * ((ks // 2) % 2)}:{143 + 4 * ((ks // 2) % 2)}], %2, s[28:31], 0 offen offset:{256 * ks}")
                    if e == "bufld_eo2" and ks % 2 == 0:
                        lines.append(f"buffer_load_dwordx2 v[{140 + 2 * ((ks // 2) % 4)}:{141 + 2 * ((ks // 2) % 4)}], %2, s[28:31], 0 offen offset:{256 * ks}")
                    if e == "bufst4_own":
                        lines.append(f"buffer_store_dwordx4 v[{144}:{147}], %7, s[28:31], 0 offen offset:{256 * (ks % 8)}")
                    if e == "bufst4_eo" and ks % 2 == 0:
                        lines.append(f"buffer_store_dwordx4 v[{144}:{147}], %7, s[28:31], 0 offen offset:{256 * (ks % 8)}")
                    if e == "bufst2_eo" and ks % 2 == 0:
                        lines.append(f"buffer_store_dwordx2 v[{144}:{145}], %7, s[28:31], 0 offen offset:{256 * (ks % 8)}")
                    if e == "bufst2_e4" and ks % 4 == 0:
                        lines.append(f"buffer_store_dwordx2 v[{144}:{145}], %7, s[28:31], 0 offen offset:{256 * (ks % 8)}")
                    if e == "wait":
                        lines.append("s_waitcnt vmcnt(8) lgkmcnt(12)")
                    if e == "salu1":
                        lines.append("s_add_u32 s26, s26, 1")
                    if e == "dsw2":
                        lines.append(f"ds_write2_b64 %2, v[{140 + 2 * (ks % 4)}:{141 + 2 * (ks % 4)}], v[{148}:{149}] offset0:{64 * (ks % 2)} offset1:{64 * (ks % 2) + 16}")
                    if e == "bufst2":
                        lines.append(f"buffer_store_dwordx2 a[{100 + 2 * (ks % 2)}:{101 + 2 * (ks % 2)}], %7, s[28:31], 0 offen offset:{256 * (ks % 8)}")
                    if e == "nop7":
                        lines.append("s_nop 7")
                    if e == "nop3":
                        lines.append("s_nop 3")
                    if e == "valu":
                        lines.append(f"v_add_u32 v{148 + (ks % 2)}, v{148 + (ks % 2)}, v150")
                    if e == "salu4":
                        lines += ["s_add_u32 s26, s26, 1"] * 4
                    if e == "bufst_own":
                        lines.append(f"buffer_store_dwordx2 v[{144 + 2 * (ks % 2)}:{145 + 2 * (ks % 2)}], %7, s[28:31], 0 offen offset:{256 * (ks % 8)}")
                    if e == "bufst_own_a":
                        lines.append(f"buffer_store_dwordx2 a[{100 + 2 * (ks % 2)}:{101 + 2 * (ks % 2)}], %7, s[28:31], 0 offen offset:{256 * (ks % 8)}")
                    if e == "ld":
                        lines.append(f"global_load_dwordx2 v[{140 + 2 * (ks % 4)}:{141 + 2 * (ks % 4)}], %3, off offset:{256 * ks}")
    return lines

out = ['// generated by tools/gen_probe_mfma4c.py -- see there', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '']
clob = ", ".join([f'"v{i}"' for i in range(10, 210)] + [f'"a{i}"' for i in range(40, 210)])
for name, (bf, cf, ex, nop, same) in VARIANTS.items():
    out.append(f'__global__ __launch_bounds__(256, 1) void k_{name}(unsigned long long* cyc, int iters, const double* src) {{')
    out.append('    __shared__ double lds[2048];')
    out.append('    lds[threadIdx.x] = 0.0; lds[threadIdx.x + 256] = 0.0; __syncthreads();')
    out.append('    unsigned lo, hi; const unsigned la = (threadIdx.x & 63) * 8; const double* p = src + (threadIdx.x & 63); const unsigned own = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096 + la;')
    out.append('    asm volatile(')
    init = []
    for r in range(10, 210):
        init.append(f"v_mov_b32 v{r}, 0")
    for r in range(40, 210):
        init.append(f"v_accvgpr_write_b32 a{r}, 0")
    init += ["s_mov_b32 s28, %5", "s_mov_b32 s29, %6", "s_mov_b32 s30, 0x800000", "s_mov_b32 s31, 0x00020000", "s_mov_b32 s20, %4", "s_memtime s[22:23]", "s_waitcnt lgkmcnt(0)", "1:"]
    tail = ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 7", "s_nop 7",
            "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s22, s24, s22", "s_subb_u32 s23, s25, s23",
            "v_mov_b32 %0, s22", "v_mov_b32 %1, s23"]
    for l in init + body(bf, cf, ex, nop, same) + tail:
        out.append(f'        "{l}\\n"')
    out.append(f'        : "=v"(lo), "=v"(hi) : "v"(la), "v"(p), "s"(iters), "s"((unsigned)(unsigned long long)src), "s"((unsigned)((unsigned long long)src >> 32) & 0xffffu), "v"(own) : "s26", "s28", "s29", "s30", "s31", "m0", "s20", "s22", "s23", "s24", "s25", "scc", "memory", {clob});')
    out.append('    if (threadIdx.x == 0) cyc[blockIdx.x] = ((unsigned long long)hi << 32) | lo;')
    out.append('}')
    out.append('')
out.append('int main() {')
out.append('    unsigned long long* d; hipMalloc(&d, 8 * 4096); double* src; hipMalloc(&src, 8 << 20); hipMemset(src, 0, 8 << 20);')
out.append('    const int iters = 2000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);')
out.append('    std::vector<unsigned long long> h(256);')
for name, (bf, cf, ex, nop, same) in VARIANTS.items():
    n = 28 if same == "pair" else 42
    out.append(f'    for (int rep = 0; rep < 2; ++rep) {{ hipEventRecord(e0); hipLaunchKernelGGL(k_{name}, dim3(256), dim3(256), 0, 0, d, iters, src); hipEventRecord(e1); hipDeviceSynchronize();')
    out.append(f'      float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h.data(), d, 8 * 256, hipMemcpyDeviceToHost);')
    out.append(f'      if (rep) printf("%-24s %6.2f cycles/MFMA (block 0)  %6.2f (block 100)   %7.2f TFLOP/s  %.3f GHz\\n", "{name}", (double)h[0] / iters / {n}, (double)h[100] / iters / {n}, 1024.0 * iters * {n} * 512 / (ms * 1e-3) / 1e12, (double)h[0] / (ms * 1e6)); }}')
out.append('    return 0;')
out.append('}')
open("tools/probe_mfma4c.hip", "w").write("\n".join(out) + "\n")
