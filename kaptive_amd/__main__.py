import os
import sys

from kaptive_amd import cli

if __name__ == "__main__":  # (the device processes of `--devices a,b` re-import this module under another name)
    # The command line ends with the process: outputs are flushed and closed by main(), and what is left -- reader threads,
    # page-locked buffers, the device context, the interpreter's own teardown -- is given back to the operating system in
    # one go instead of being unwound (0.6 s + 0.4 s on a run that types 18 000 assemblies in 2 s).
    # Only `assembly` / `type` ends that way, and only after run_type has closed its own output handles
    # (cli.FAST_EXIT_ARMED); every other subcommand -- and a run that failed -- leaves through the interpreter, with its exit
    # handlers and buffered streams.  The kernel still unpins those pages and takes the context apart after the parent has
    # seen the process end: a command started right behind a large run pays for that (bench.py's CLI leg waits 2 s, untimed,
    # and says so in its line).
    cli.FAST_EXIT = True
    rc = cli.main()
    rc = rc if isinstance(rc, int) else 1
    sys.stdout.flush()
    sys.stderr.flush()
    if cli.FAST_EXIT_ARMED:
        os._exit(rc)
    sys.exit(rc)
