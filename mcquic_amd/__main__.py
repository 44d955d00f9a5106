import sys

from .demo import main

sys.exit(main())
