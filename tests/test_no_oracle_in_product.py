"""The oracle and the wave emulator are test infrastructure: nothing under dial_mpc_amd/ (nor bench.py's
timed path) may import or load them."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_package_never_references_oracle_or_emulator():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dial_mpc_amd")):
        for f in files:
            if not f.endswith((".py", ".h", ".hip", ".cpp")):
                continue
            text = open(os.path.join(dirpath, f)).read()
            for pat in (r"import\s+oracle", r"from\s+oracle", r"liboracle", r"libwave_emu", r"emu_lib",
                        r"oracle/dial_oracle"):
                if re.search(pat, text):
                    bad.append((os.path.join(dirpath, f), pat))
    assert not bad, bad


def test_bench_uses_oracle_only_in_cpu_baseline():
    text = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"oracle", text)]
    body = text[text.index("def cpu_baseline"):text.index("def ", text.index("def cpu_baseline") + 10)]
    outside = re.sub(re.escape(body), "", text)
    assert uses and not re.search(r"import\s+oracle|liboracle", outside)
