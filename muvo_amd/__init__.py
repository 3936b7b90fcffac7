"""MI355X-native MUVO world-model training step."""
import os

# The step runs independent sub-networks on a few HIP streams (muvo_amd/ops.py: side_stream).  With the runtime's default of four
# hardware queues per process, streams of other libraries (RCCL) push them onto shared queues in an order nobody controls; eight
# queues give every stream its own.  Read by the HIP runtime when it initialises, i.e. at the first GPU call - importing this
# package before touching the GPU is enough.  An explicit setting in the environment wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
