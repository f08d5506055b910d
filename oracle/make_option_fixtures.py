"""TEST INFRASTRUCTURE — dumps the shipped option files of the reference (ssr/options/*.yml, parsed with PyYAML) to
tests/golden/ssr_options.json so that the option-handling tests can run where /root/reference does not exist.
Configuration data only (no code is copied).  Run in the build container:  python oracle/make_option_fixtures.py"""
import glob
import json
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ssr/options"


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "*.yml"))):
        with open(path) as f:
            out[os.path.basename(path)] = yaml.safe_load(f)
    dst = os.path.join(ROOT, "tests", "golden", "ssr_options.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst, sorted(out))


if __name__ == "__main__":
    main()
