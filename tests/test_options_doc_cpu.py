"""The option surface is ONE list: the `-hip_*` options INTEGRATION.md documents == the ones the library's own
help prints == the ones apply_command_line_options accepts -- and there are at most twelve of them."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _names(text):
    return set(re.findall(r"-(?:\[no-\])?(hip_[a-z_]+)", text))


def test_documented_options_equal_help_equal_parser():
    src = (ROOT / "yask_amd" / "csrc" / "ykh_solution.cpp").read_text()
    help_txt = src[src.index("std::string Solution::get_command_line_help()"):src.index("std::string Solution::get_command_line_values()")]
    parser = src[src.index("std::string Solution::apply_command_line_options("):src.index("std::string Solution::get_command_line_help()")]
    tables = "".join(re.findall(r'const char\* (?:bool_opts|int_opts|dbl_opts|str_opts)\[\] = \{(.*?)\};', parser, re.S))
    accepted = set(re.findall(r'"(hip_[a-z_]+)"', tables))
    in_help = _names(help_txt)
    doc = (ROOT / "INTEGRATION.md").read_text()
    table = doc[doc.index("### Options and environment variables of this library"):]
    documented = _names(table[:table.index("| environment variable")])
    assert accepted == in_help == documented, (accepted ^ in_help, accepted ^ documented)
    assert len(accepted) <= 12, sorted(accepted)
    env_doc = set(re.findall(r"`(YASK_HIP_[A-Z_]+)", table))
    env_src = set()
    for f in list((ROOT / "yask_amd" / "csrc").glob("*.cpp")) + [ROOT / "yask_amd" / "_capi.py"]:
        env_src |= set(re.findall(r'"(YASK_HIP_[A-Z_]+)"', f.read_text()))
    assert env_src == env_doc, env_src ^ env_doc
