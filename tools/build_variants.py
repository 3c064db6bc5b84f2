"""Build A/B variants of libharl_hip.so (harl_amd/lib/libharl_<name>.so, selected at run time with HARL_LIB=<path>).

    python tools/build_variants.py sv2=-DHARL_SPLIT_VARIANT=2 sv3=-DHARL_SPLIT_VARIANT=3

Every variant gets the library's default per-file flags plus the given -D switches on every translation unit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harl_amd import _build  # noqa: E402


def main():
    for item in sys.argv[1:]:
        name, _, flags = item.partition("=")
        add = flags.split(",") if flags else []
        extra = {src: list(_build.EXTRA_FLAGS.get(src, [])) + add for src in _build.SOURCES}
        print(name, _build.build(force=False, variant=name, extra=extra), flush=True)


if __name__ == "__main__":
    main()
