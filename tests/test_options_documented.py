"""Every option ttcr_fsm_set_option accepts (GridBase::apply_option, ttcr_amd/csrc/fsm_capi.hip) is described in include/ttcr_amd.h, and
the header describes no option the library would refuse."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_library_name_the_same_options():
    src = open(os.path.join(ROOT, "ttcr_amd", "csrc", "fsm_capi.hip")).read()
    body = src[src.index("void apply_option(const std::string& k, double value) {"):]
    body = body[:body.index("unknown option")]
    accepted = set(re.findall(r'k == "([a-z_]+)"', body))
    hdr = open(os.path.join(ROOT, "include", "ttcr_amd.h")).read()
    documented = set(re.findall(r'^ \*   "([a-z_]+)"', hdr, flags=re.M))
    assert accepted and accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))
