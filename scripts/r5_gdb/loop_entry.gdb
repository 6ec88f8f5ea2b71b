set pagination off
set confirm off
set breakpoint pending on
set height 0
break cdkf_custom_kernel
run
delete 1
break *(&cdkf_custom_kernel + 0x4d0)
commands
silent
printf "LOOPTEST T(s27:s26)=%#x:%#x (T<1)(s1:s0)=%#x:%#x vcc=%#lx exec=%#lx vccz=%d\n", $s27, $s26, $s1, $s0, $vcc, $exec, ($status >> 9) & 1
continue
end
break *(&cdkf_custom_kernel + 0x2c6f0)
commands
silent
printf "ZERO-T block entered\n"
continue
end
break *(&cdkf_custom_kernel + 0x504)
commands
silent
printf "LOOP path entered\n"
continue
end
continue
quit
