import sys

from kaptive_amd.cli import main

sys.exit(main())
