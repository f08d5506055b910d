"""One naming rule for kernel symbols across bench.py (roofline.kernel), tools/pmc_traffic.py, tools/pmc_sq.py and
tools/roofline_check.py: the name rocprofv3 prints, without the "void (anonymous namespace)::" prefix and the argument list; the
straight-line epilogue variants of one template (conv_x3r_kernel<NT, EP>) fold into their FAMILY conv_x3r_kernel<NT>."""
import re


def bare(name: str) -> str:
    """rocprofv3 Kernel_Name / Name -> template name with its argument list stripped"""
    s = name.strip().strip('"')
    s = re.sub(r"^void\s+", "", s)
    s = s.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in s:                      # cut at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def family(name: str) -> str:
    s = bare(name)
    m = re.match(r"conv_x3r_kernel<(\d), (\d), \d, (\d), (\w+)>", s)      # <channel tiles per wave, wave groups, epilogue, tile height, arithmetic>
    if m:      # arithmetic 0 (split-bf16) and 2 (split-fp16, the forward launches of mode fp32h): one row, as in bench.py family()
        am = "0|2" if m.group(4) in ("0", "2") else m.group(4)
        return f"conv_x3r_kernel<{m.group(1)}, {m.group(2)}, *, {m.group(3)}, {am}>"
    return s


HOT = re.compile(r"^(rdbt_kernel|rdb_kernel|wgrad_|conv_|split_bf16_multi_kernel|bilinear2x_)")


def hot(name: str):
    """family name of a kernel of this library's conv / wgrad / dense-block groups, else None (ATen fills, copies, ...)"""
    f = family(name)
    return f if HOT.match(f) else None
