import sys
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
sys.argv = sys.argv[:1]
exec(open("tools/explore_saw8.py").read())
