"""Own implementation of the design API.  The modules are deliberately importable by their bare names
(`from DesignCSG import *`, `import scenecompiler`, `from designlibrary import *`) with this directory on
sys.path, exactly like the reference's working directory, so existing design scripts run unchanged:
see designcsg_amd.run."""
