bash tools/_run3.sh
VARIANTS="abl1 abl3" bash tools/_run2.sh
