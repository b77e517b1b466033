#!/bin/bash
# five back-to-back runs of the driver's bench command (CPU baseline skipped after the first): run-to-run spread of the headline
for i in 1 2 3 4 5; do
  python bench.py --steps 20 --warmup 5 $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > /tmp/rep_$i.json
  python3 -c "
import json; d=json.load(open('/tmp/rep_$i.json')); o=d['config']['other_frame_contents']; s=d['config']['other_launch_model']
print('run $i: value %.0f fps  frac %.4f  avg_launch %.2f us | natural %.4f  random %.4f | streams %.0f fps = %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms']*1e3, o['natural']['frac'], o['random']['frac'], s['value'], s['frac']))"
done
