import torch
print('range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for pr in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=pr); print(pr,'->',s.priority)
    except Exception as e: print(pr,'err',e)
print('default', torch.cuda.current_stream().priority)
