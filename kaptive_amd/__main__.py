import sys

from kaptive_amd.cli import main

if __name__ == "__main__":  # (the device processes of `--devices a,b` re-import this module under another name)
    sys.exit(main())
